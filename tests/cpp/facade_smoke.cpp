// Builds against the C++ facade (reference class / method names) and libsvo_hip.so.
// With a GPU: one tiny stereo pair through StereoSlam::new_image, prints the keypoint count.
// Without: the library must fail loudly (exception text from svo_last_error), exit code 3.
#include <cstdio>
#include <vector>

#include "../../stereo-svo-slam_amd/hostcpp/stereo_slam.hpp"

int main() {
    svo_amd::CameraSettings cam{};
    cam.baseline = 20.f; cam.fx = 200.f; cam.fy = 200.f; cam.cx = 160.f; cam.cy = 120.f;
    cam.grid_height = 40; cam.grid_width = 40; cam.search_x = 30; cam.search_y = 4;
    cam.window_size_pose_estimator = 4; cam.window_size_opt_flow = 21;
    cam.window_size_depth_calculator = 21; cam.max_pyramid_levels = 4;
    cam.min_pyramid_level_pose_estimation = 1;
    const int w = 320, h = 240;
    std::vector<uint8_t> img((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)y * w + x] = (uint8_t)(((x / 8) ^ (y / 8)) & 1 ? 200 : 40);
    try {
        svo_amd::StereoSlam slam(cam);
        const svo_amd::Image8 view{img.data(), w, h, w};
        slam.new_image(view, view, 0.f);
        svo_amd::Frame f;
        if (!slam.get_frame(f)) return 4;
        std::printf("keypoints %zu\n", f.kps.kps2d.size());
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
    return 0;
}
