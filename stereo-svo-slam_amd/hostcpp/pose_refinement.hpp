// pose_refinement.hpp — PoseRefiner (src/include/pose_refinement.hpp:22-32) over the C ABI:
// KLT alignment of every keypoint against its origin keyframe, the merge rules and the
// reprojection Gauss-Newton (src/lib/pose_refinement.cpp:62-177, 236-290).
#pragma once

#include <map>

#include "optical_flow.hpp"

namespace svo_amd {

class PoseRefiner {
public:
    PoseRefiner(const Handle& handle, const CameraSettings& camera_settings) : h(handle), camera_settings(camera_settings) {}

    // float refine_pose(KeyFrameManager& keyframe_manager, Frame& frame): frame.kps.kps2d holds the
    // projections at frame.pose on entry (StereoSlam::estimate_pose, src/lib/stereo_slam.cpp:73-80);
    // on return the tracked positions / flags and the refined frame.pose. Returns the final cost.
    float refine_pose(KeyFrameManager& keyframe_manager, Frame& frame) {
        KeyPoints& kps = frame.kps;
        const int n = (int)kps.kps2d.size();
        // split by origin keyframe, ascending id (std::map, pose_refinement.cpp:72-82)
        std::map<int, std::vector<int>> groups;
        for (int i = 0; i < n; i++) groups[kps.info[i].keyframe_id].push_back(i);
        std::vector<KeyPoint2d> tracked(n);
        std::vector<float> err(n, 0.f);
        OpticalFlow optical_flow(h, camera_settings);
        for (auto& g : groups) {
            KeyFrame* keyframe = keyframe_manager.get_keyframe((uint32_t)g.first);
            if (!keyframe) throw std::runtime_error("refine_pose: unknown keyframe id");
            std::vector<KeyPoint2d> ref, cur;
            for (int i : g.second) {
                ref.push_back(keyframe->kps.kps2d[kps.info[i].keypoint_index]);
                cur.push_back(kps.kps2d[i]);
            }
            std::vector<float> e;
            optical_flow.calculate_optical_flow(keyframe->stereo_image, ref, frame.stereo_image, cur, e);
            for (size_t k = 0; k < g.second.size(); k++) {
                tracked[g.second[k]] = cur[k];
                err[g.second[k]] = e[k];
            }
        }
        // merge (:125-150) + update_pose (:236-290) in one launch of reproj_gn_kernel
        std::vector<uint32_t> fl(n);
        for (int i = 0; i < n; i++) fl[i] = flags_of(kps.info[i]);
        DeviceArray<KeyPoint2d> d2(n), dtr(n);
        DeviceArray<KeyPoint3d> d3(n);
        DeviceArray<uint32_t> df(n);
        DeviceArray<float> derr(n), dpose(16);
        d2.upload(h, kps.kps2d.data(), n);
        dtr.upload(h, tracked.data(), n);
        d3.upload(h, kps.kps3d.data(), n);
        df.upload(h, fl.data(), n);
        derr.upload(h, err.data(), n);
        const Vec6f p = frame.pose.get_vector();
        dpose.upload(h, p.data(), 6);
        check(svo_reproj_gn(h.get(), d2.data(), d3.data(), df.data(), n, &camera_settings, dtr.data(),
                            derr.data(), dpose.data(), dpose.data() + 6, dpose.data() + 12, nullptr));
        d2.download(h, kps.kps2d.data(), n);
        df.download(h, fl.data(), n);
        for (int i = 0; i < n; i++) set_flags(kps.info[i], fl[i]);
        float out[7];
        check(svo_copy_to_host(h.get(), out, dpose.data() + 6, sizeof(out)));
        frame.pose.set_vector({out[0], out[1], out[2], out[3], out[4], out[5]});
        return out[6];
    }

private:
    const Handle& h;
    const CameraSettings camera_settings;
};

}  // namespace svo_amd
