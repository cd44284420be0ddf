"""stereo_svo_slam_amd — MI355X-native hot path of the stereo SVO library
(sparse image alignment, KLT refinement, stereo depth filter) behind the
reference's StereoSlam / PoseEstimator surface. See DESIGN.md."""
