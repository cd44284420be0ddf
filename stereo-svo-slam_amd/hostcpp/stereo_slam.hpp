// stereo_slam.hpp — C++ facade with the reference's names over the C ABI
// (include/svo_hip.h). Mirrors src/include/stereo_slam.hpp:27-79 and the POD
// types of src/include/stereo_slam_types.hpp. Images are passed as plain
// 8-bit views; when OpenCV is available the cv::Mat overloads below compile
// too, so src/app/slam_app.cpp (SlamApp::process_image, :160-196) links
// against this header unchanged.
#pragma once

#include "stereo_slam_types.hpp"

namespace svo_amd {

class StereoSlam {
public:
    explicit StereoSlam(const CameraSettings& camera_settings, int device = 0)
        : camera_settings(camera_settings), device(device) {}
    ~StereoSlam() { if (ctx) svo_ctx_destroy(ctx); }
    StereoSlam(const StereoSlam&) = delete;
    StereoSlam& operator=(const StereoSlam&) = delete;

    // svo_ctx_set_fast_solver: off (default) = the reference's Gauss-Newton arithmetic, bit for bit
    void set_fast_solver(bool on) { fast = on; if (ctx) check(svo_ctx_set_fast_solver(ctx, on ? 1 : 0)); }

    // void new_image(const cv::Mat& left, const cv::Mat& right, const float time_stamp)
    void new_image(const Image8& left, const Image8& right, const float time_stamp) {
        if (!ctx) {
            check(svo_ctx_create(&camera_settings, left.cols, left.rows, 1, device, &ctx));
            if (fast) check(svo_ctx_set_fast_solver(ctx, 1));
        }
        check(svo_new_image(ctx, left.data, left.step, right.data, right.step, left.cols, left.rows,
                            time_stamp));
        last_ts = time_stamp;
    }
#ifdef SVO_FACADE_HAVE_OPENCV
    void new_image(const cv::Mat& left, const cv::Mat& right, const float time_stamp) {
        new_image(view_of(left), view_of(right), time_stamp);
    }
#endif

    bool get_frame(Frame& frame) {
        if (!ctx) return false;                       // false before the first image (:278-284)
        int n = 0;
        check(svo_get_frame_keypoints(ctx, 0, nullptr, nullptr, nullptr, 0, &n));
        frame.kps.kps2d.resize(n); frame.kps.kps3d.resize(n); frame.kps.info.resize(n);
        check(svo_get_frame_keypoints(ctx, 0, frame.kps.kps2d.data(), frame.kps.kps3d.data(),
                                      frame.kps.info.data(), n, &n));
        Pose p{};
        check(svo_get_pose(ctx, 0, &p.x));
        frame.pose.set_pose(p);
        svo_frame_stats st;
        check(svo_get_frame_stats(ctx, 0, &st));
        frame.id = (uint64_t)st.frame_id;
        frame.time_stamp = last_ts;
        return true;
    }
    void get_keyframe(KeyFrame& keyframe) {
        int count = 0;
        check(svo_get_keyframe_count(ctx, 0, &count));
        read_keyframe(count - 1, keyframe);
    }
    void get_keyframes(std::vector<KeyFrame>& keyframes) {
        int count = 0;
        check(svo_get_keyframe_count(ctx, 0, &count));
        keyframes.resize(count);
        for (int i = 0; i < count; i++) read_keyframe(i, keyframes[i]);
    }
    void get_trajectory(std::vector<Pose>& trajectory) {
        int n = 0;
        check(svo_get_trajectory(ctx, 0, nullptr, 0, &n));
        trajectory.resize(n);
        check(svo_get_trajectory(ctx, 0, trajectory.data(), n, &n));
    }
    // Pose update_pose(const Pose&, const cv::Vec6f& speed, const cv::Vec6f& pose_variance,
    //                  const cv::Vec6f& speed_variance, double dt)
    Pose update_pose(const Pose& pose, const float speed[6], const float pose_variance[6],
                     const float speed_variance[6], double dt) {
        Pose out{};
        check(svo_update_pose(ctx, 0, &pose.x, speed, pose_variance, speed_variance, dt, &out.x));
        return out;
    }

private:
    void read_keyframe(int id, KeyFrame& kf) {
        int n = 0;
        Pose p{};
        check(svo_get_keyframe(ctx, 0, id, nullptr, nullptr, nullptr, &p.x, 0, &n));
        kf.kps.kps2d.resize(n); kf.kps.kps3d.resize(n); kf.kps.info.resize(n);
        check(svo_get_keyframe(ctx, 0, id, kf.kps.kps2d.data(), kf.kps.kps3d.data(),
                               kf.kps.info.data(), &p.x, n, &n));
        kf.pose.set_pose(p);
        kf.id = (uint64_t)id;
    }
    const CameraSettings camera_settings;
    int device;
    svo_ctx* ctx = nullptr;
    double last_ts = 0;
    bool fast = false;
};

}  // namespace svo_amd
