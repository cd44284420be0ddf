"""stereo_svo_slam_amd — MI355X-native hot path of the stereo SVO library
(sparse image alignment, KLT refinement, stereo depth filter) behind the
reference's StereoSlam / PoseEstimator surface. See DESIGN.md.

Importing the package changes nothing in the process environment. A ctx with more than three
sequence groups needs more hardware queues than the HIP runtime's default of four
(GPU_MAX_HW_QUEUES, read by the runtime when it initialises): the application sets it, as
bench.py does, before anything touches the GPU."""
