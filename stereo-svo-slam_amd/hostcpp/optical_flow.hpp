// optical_flow.hpp — OpticalFlow (src/include/optical_flow.hpp:26-30) over the C ABI:
// cv::calcOpticalFlowPyrLK as src/lib/optical_flow.cpp:14-56 calls it (3 levels, window
// window_size_opt_flow, 30 iterations / 0.01, initial flow from current_keypoints2d, +inf error
// for lost points), one launch of klt_track_kernel; project_keypoints
// (src/include/transform_keypoints.hpp:17-19).
#pragma once

#include "stereo_slam_types.hpp"

namespace svo_amd {

// void project_keypoints(const PoseManager&, const std::vector<KeyPoint3d>& in, const CameraSettings&,
//                        std::vector<KeyPoint2d>& out)
inline void project_keypoints(const Handle& h, const PoseManager& pose, const std::vector<KeyPoint3d>& in,
                              const CameraSettings& camera_settings, std::vector<KeyPoint2d>& out) {
    const int n = (int)in.size();
    out.resize(n);
    DeviceArray<KeyPoint3d> d3(n);
    DeviceArray<KeyPoint2d> d2(n);
    DeviceArray<float> dp(6);
    d3.upload(h, in.data(), n);
    const Vec6f v = pose.get_vector();
    dp.upload(h, v.data(), 6);
    check(svo_project_keypoints(h.get(), dp.data(), d3.data(), n, &camera_settings, d2.data()));
    d2.download(h, out.data(), n);
}

class OpticalFlow {
public:
    OpticalFlow(const Handle& handle, const CameraSettings& camera_settings) : h(handle), camera_settings(camera_settings) {}

    void calculate_optical_flow(const StereoImage& previous_stereo_image_pyr,
                                const std::vector<KeyPoint2d>& previous_keypoints2d,
                                const StereoImage& current_stereo_image_pyr,
                                std::vector<KeyPoint2d>& current_keypoints2d, std::vector<float>& err) {
        const int n = (int)previous_keypoints2d.size();
        err.assign(n, 0.f);
        if (n == 0) return;
        DeviceArray<KeyPoint2d> dprev(n), dcur(n);
        DeviceArray<float> derr(n);
        DeviceArray<uint8_t> dst(n);
        dprev.upload(h, previous_keypoints2d.data(), n);
        dcur.upload(h, current_keypoints2d.data(), n);
        const int nl = (int)std::min(previous_stereo_image_pyr.opt_flow.size(), current_stereo_image_pyr.opt_flow.size());
        std::vector<svo_image> pl(nl), cl(nl);
        for (int l = 0; l < nl; l++) {
            pl[l] = previous_stereo_image_pyr.opt_flow[l].view;
            cl[l] = current_stereo_image_pyr.opt_flow[l].view;
        }
        check(svo_klt_track(h.get(), pl.data(), cl.data(), nl, dprev.data(), dcur.data(), n,
                            camera_settings.window_size_opt_flow, dst.data(), derr.data()));
        dcur.download(h, current_keypoints2d.data(), n);
        derr.download(h, err.data(), n);       // +inf where status == 0 (optical_flow.cpp:46-50)
    }

private:
    const Handle& h;
    const CameraSettings camera_settings;
};

}  // namespace svo_amd
