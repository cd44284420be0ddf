// depth_filter.hpp — DepthFilter (src/include/depth_filter.hpp:14-20) over the C ABI:
// calculate_disparities (ssd_disparity_kernel), outlier_check and update_kps3d
// (filter_update_kernel) of src/lib/depth_filter.cpp:40-50, 52-128, 130-257, 259-327.
#pragma once

#include "stereo_slam_types.hpp"

namespace svo_amd {

class DepthFilter {
public:
    DepthFilter(const Handle& handle, KeyFrameManager& keyframe_manager, const CameraSettings& camera_settings)
        : h(handle), keyframe_manager(keyframe_manager), camera_settings(camera_settings) {}

    // void update_depth(Frame& frame, std::vector<KeyPoint3d>& updated_kps3d): like the reference it
    // bumps the inlier / outlier counters and the per-point filter state in frame.kps.info and
    // returns the new points in updated_kps3d (the caller writes them back, stereo_slam.cpp:205-229)
    void update_depth(Frame& frame, std::vector<KeyPoint3d>& updated_kps3d) {
        KeyPoints& kps = frame.kps;
        const int n = (int)kps.kps2d.size();
        updated_kps3d = kps.kps3d;
        if (n == 0) return;
        std::vector<KeyPoint3d> ref3d(n);
        std::vector<KeyPoint2d> ref2d(n);
        std::vector<float> kf_pose((size_t)n * 6), kfx(n), kfP(n);
        std::vector<uint32_t> fl(n);
        std::vector<int32_t> outl(n), inl(n);
        for (int i = 0; i < n; i++) {
            const KeyPointInformation& info = kps.info[i];
            KeyFrame* keyframe = keyframe_manager.get_keyframe((uint32_t)info.keyframe_id);
            if (!keyframe) throw std::runtime_error("update_depth: unknown keyframe id");
            ref3d[i] = keyframe->kps.kps3d[info.keypoint_index];
            ref2d[i] = keyframe->kps.kps2d[info.keypoint_index];
            const Vec6f kp = keyframe->pose.get_vector();
            for (int j = 0; j < 6; j++) kf_pose[(size_t)i * 6 + j] = kp[j];
            fl[i] = flags_of(info);
            outl[i] = info.outlier_count; inl[i] = info.inlier_count;
            kfx[i] = info.kf_inv_depth; kfP[i] = info.kf_variance;
        }
        DeviceArray<KeyPoint2d> d2(n), dref2(n);
        DeviceArray<KeyPoint3d> d3(n), dref3(n);
        DeviceArray<uint32_t> df(n);
        DeviceArray<int32_t> dout(n), din(n);
        DeviceArray<float> ddisp(n), dkfpose((size_t)n * 6), dkfx(n), dkfP(n), dpose(6);
        d2.upload(h, kps.kps2d.data(), n); d3.upload(h, kps.kps3d.data(), n);
        dref2.upload(h, ref2d.data(), n); dref3.upload(h, ref3d.data(), n);
        df.upload(h, fl.data(), n); dout.upload(h, outl.data(), n); din.upload(h, inl.data(), n);
        dkfpose.upload(h, kf_pose.data(), (size_t)n * 6); dkfx.upload(h, kfx.data(), n); dkfP.upload(h, kfP.data(), n);
        const Vec6f fp = frame.pose.get_vector();
        dpose.upload(h, fp.data(), 6);
        check(svo_ssd_disparity(h.get(), &frame.stereo_image.left[0].view, &frame.stereo_image.right[0].view,
                                d2.data(), n, camera_settings.window_size_depth_calculator,
                                camera_settings.search_x, camera_settings.search_y, 1, ddisp.data()));
        check(svo_depth_filter_update(h.get(), d2.data(), d3.data(), df.data(), n, &camera_settings, dpose.data(),
                                      ddisp.data(), dref3.data(), dref2.data(), dkfpose.data(), dout.data(),
                                      din.data(), dkfx.data(), dkfP.data(), 1, 1));
        d3.download(h, updated_kps3d.data(), n);
        dout.download(h, outl.data(), n); din.download(h, inl.data(), n);
        dkfx.download(h, kfx.data(), n); dkfP.download(h, kfP.data(), n);
        for (int i = 0; i < n; i++) {
            kps.info[i].outlier_count = outl[i]; kps.info[i].inlier_count = inl[i];
            kps.info[i].kf_inv_depth = kfx[i]; kps.info[i].kf_variance = kfP[i];
        }
    }

private:
    const Handle& h;
    KeyFrameManager& keyframe_manager;
    const CameraSettings camera_settings;
};

}  // namespace svo_amd
