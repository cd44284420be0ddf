// stereo_slam.hpp — C++ facade with the reference's names over the C ABI
// (include/svo_hip.h). Mirrors src/include/stereo_slam.hpp:27-79 and the POD
// types of src/include/stereo_slam_types.hpp. Images are passed as plain
// 8-bit views; when OpenCV is available the cv::Mat overloads below compile
// too, so src/app/slam_app.cpp (SlamApp::process_image, :160-196) links
// against this header unchanged.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/svo_hip.h"

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define SVO_FACADE_HAVE_OPENCV 1
#endif
#endif

namespace svo_amd {

using CameraSettings = svo_camera_settings;   // same field order as the reference
using KeyPoint2d = svo_kp2d;
using KeyPoint3d = svo_kp3d;
using Pose = svo_pose;
using KeyPointInformation = svo_kp_info;

struct KeyPoints {
    std::vector<KeyPoint2d> kps2d;
    std::vector<KeyPoint3d> kps3d;
    std::vector<KeyPointInformation> info;
};

struct Frame {
    uint64_t id = 0;
    Pose pose{};
    KeyPoints kps;
    double time_stamp = 0;
};
struct KeyFrame : Frame {};

struct Image8 {            // CV_8U single channel view
    const uint8_t* data;
    int cols, rows, step;
};

class StereoSlam {
public:
    explicit StereoSlam(const CameraSettings& camera_settings, int device = 0)
        : camera_settings(camera_settings), device(device) {}
    ~StereoSlam() { if (ctx) svo_ctx_destroy(ctx); }
    StereoSlam(const StereoSlam&) = delete;
    StereoSlam& operator=(const StereoSlam&) = delete;

    // void new_image(const cv::Mat& left, const cv::Mat& right, const float time_stamp)
    void new_image(const Image8& left, const Image8& right, const float time_stamp) {
        if (!ctx) check(svo_ctx_create(&camera_settings, left.cols, left.rows, 1, device, &ctx));
        check(svo_new_image(ctx, left.data, left.step, right.data, right.step, left.cols, left.rows,
                            time_stamp));
        last_ts = time_stamp;
    }
#ifdef SVO_FACADE_HAVE_OPENCV
    void new_image(const cv::Mat& left, const cv::Mat& right, const float time_stamp) {
        CV_Assert(left.type() == CV_8U && right.type() == CV_8U);
        new_image(Image8{left.data, left.cols, left.rows, (int)left.step},
                  Image8{right.data, right.cols, right.rows, (int)right.step}, time_stamp);
    }
#endif

    bool get_frame(Frame& frame) {
        if (!ctx) return false;                       // false before the first image (:278-284)
        int n = 0;
        check(svo_get_frame_keypoints(ctx, 0, nullptr, nullptr, nullptr, 0, &n));
        frame.kps.kps2d.resize(n); frame.kps.kps3d.resize(n); frame.kps.info.resize(n);
        check(svo_get_frame_keypoints(ctx, 0, frame.kps.kps2d.data(), frame.kps.kps3d.data(),
                                      frame.kps.info.data(), n, &n));
        check(svo_get_pose(ctx, 0, &frame.pose.x));
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
        check(svo_get_keyframe(ctx, 0, id, nullptr, nullptr, nullptr, &kf.pose.x, 0, &n));
        kf.kps.kps2d.resize(n); kf.kps.kps3d.resize(n); kf.kps.info.resize(n);
        check(svo_get_keyframe(ctx, 0, id, kf.kps.kps2d.data(), kf.kps.kps3d.data(),
                               kf.kps.info.data(), &kf.pose.x, n, &n));
        kf.id = (uint64_t)id;
    }
    // the reference's methods return void and print diagnostics; a failing HIP call has no
    // analogue there, so it is surfaced as an exception instead of being swallowed
    static void check(int rc) {
        if (rc != SVO_OK) throw std::runtime_error(std::string("libsvo_hip: ") + svo_last_error());
    }
    const CameraSettings camera_settings;
    int device;
    svo_ctx* ctx = nullptr;
    double last_ts = 0;
};

}  // namespace svo_amd
