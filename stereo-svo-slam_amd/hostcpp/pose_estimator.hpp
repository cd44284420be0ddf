// pose_estimator.hpp — PoseEstimator (src/include/pose_estimator.hpp:19-27) over the C ABI:
// sparse image alignment of the current frame against the previous one, coarse to fine
// (src/lib/pose_estimator.cpp:115-130), one launch of sia_prep_kernel + sia_gn_kernel.
#pragma once

#include "stereo_slam_types.hpp"

namespace svo_amd {

class PoseEstimator {
public:
    // PoseEstimator(const StereoImage& current, const StereoImage& previous,
    //               const KeyPoints& previous_keypoints, const CameraSettings&)
    PoseEstimator(const Handle& handle, const StereoImage& current_stereo_image,
                  const StereoImage& previous_stereo_image, const KeyPoints& previous_keypoints,
                  const CameraSettings& camera_settings)
        : h(handle), cur(current_stereo_image), prev(previous_stereo_image), kps(previous_keypoints),
          camera_settings(camera_settings) {}

    // float estimate_pose(const PoseManager& pose_manager_guess, PoseManager& estimated_pose)
    float estimate_pose(const PoseManager& pose_manager_guess, PoseManager& estimated_pose) {
        const int n = (int)kps.kps2d.size();
        std::vector<uint32_t> fl(n);
        for (int i = 0; i < n; i++) fl[i] = flags_of(kps.info[i]);
        DeviceArray<KeyPoint2d> d2(n);
        DeviceArray<KeyPoint3d> d3(n);
        DeviceArray<uint32_t> df(n);
        DeviceArray<float> dpose(16);
        d2.upload(h, kps.kps2d.data(), n);
        d3.upload(h, kps.kps3d.data(), n);
        df.upload(h, fl.data(), n);
        const Vec6f g = pose_manager_guess.get_vector();
        dpose.upload(h, g.data(), 6);
        std::vector<svo_image> pv(SVO_MAX_PYRAMID_LEVELS), cv_(SVO_MAX_PYRAMID_LEVELS);
        for (int l = 0; l < camera_settings.max_pyramid_levels; l++) {
            pv[l] = prev.left[l].view;
            cv_[l] = cur.left[l].view;
        }
        check(svo_sparse_align(h.get(), pv.data(), cv_.data(), d2.data(), d3.data(), df.data(), n,
                               &camera_settings, dpose.data(), dpose.data() + 6, dpose.data() + 12, nullptr,
                               nullptr, -1));
        float out[7];
        check(svo_copy_to_host(h.get(), out, dpose.data() + 6, sizeof(out)));
        estimated_pose.set_vector({out[0], out[1], out[2], out[3], out[4], out[5]});
        return out[6];
    }

private:
    const Handle& h;
    const StereoImage& cur;
    const StereoImage& prev;
    const KeyPoints& kps;
    const CameraSettings camera_settings;
};

}  // namespace svo_amd
