"""stereo_svo_slam_amd — MI355X-native hot path of the stereo SVO library
(sparse image alignment, KLT refinement, stereo depth filter) behind the
reference's StereoSlam / PoseEstimator surface. See DESIGN.md."""
import os as _os

# A ctx with more than three sequence groups needs more than the HIP runtime's default of four
# hardware queues (streams that share a queue serialise). Only effective before HIP initialises;
# an explicit setting of the caller wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
