// stereo_slam_types.hpp — the reference's data types (src/include/stereo_slam_types.hpp:16-131,
// src/include/pose_manager.hpp:21-63) for the C++ facades over the C ABI of libsvo_hip.so.
// POD types are the C ABI's own (same field order as the reference); images live in device
// memory behind a small owning view, because the classes that take a StereoImage
// (PoseEstimator, PoseRefiner, OpticalFlow, DepthFilter) run on the GPU. No HIP headers are
// needed to build against this: device memory goes through svo_device_malloc / svo_copy_*.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/svo_hip.h"
#include "../../include/svo_libm.h"

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
using Matx33f = std::array<float, 9>;
using Vec3f = std::array<float, 3>;
using Vec6f = std::array<float, 6>;

// the reference's methods return void and print diagnostics; a failing HIP call has no
// analogue there, so it is surfaced as an exception instead of being swallowed
inline void check(int rc) {
    if (rc != SVO_OK) throw std::runtime_error(std::string("libsvo_hip: ") + svo_last_error());
}

// ---- PoseManager (src/include/pose_manager.hpp:40-63, src/lib/pose_manager.cpp:9-80):
// caches R(r) and R(-r); cv::Rodrigues restated in double with the shared sin / cos
class PoseManager {
public:
    PoseManager() { Pose p{}; set_pose(p); }
    void set_pose(const Pose& p) {
        pose = p;
        angles = {p.rx, p.ry, p.rz};
        translation = {p.x, p.y, p.z};
        double R[9];
        rodrigues(angles.data(), R);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                rot_mat[i * 3 + j] = (float)R[i * 3 + j];
                inv_rot_mat[i * 3 + j] = (float)R[j * 3 + i];    // R(-r) is the transpose
            }
    }
    void set_vector(const Vec6f& v) { set_pose(Pose{v[0], v[1], v[2], v[3], v[4], v[5]}); }
    Matx33f get_rotation_matrix() const { return rot_mat; }
    Matx33f get_inv_rotation_matrix() const { return inv_rot_mat; }
    Vec3f get_translation() const { return translation; }
    Vec3f get_angles() const { return angles; }
    // rotation applied in inverse order, R_z (R_x R_y) (pose_manager.cpp:46-60)
    Vec3f get_robot_angles() const {
        double Rd[9];
        float fx[9], fy[9], fz[9], ft[9], fm[9];
        const float ax[3] = {angles[0], 0, 0}, ay[3] = {0, angles[1], 0}, az[3] = {0, 0, angles[2]};
        rodrigues(ax, Rd); for (int i = 0; i < 9; i++) fx[i] = (float)Rd[i];
        rodrigues(ay, Rd); for (int i = 0; i < 9; i++) fy[i] = (float)Rd[i];
        rodrigues(az, Rd); for (int i = 0; i < 9; i++) fz[i] = (float)Rd[i];
        mul33(fx, fy, ft);
        mul33(fz, ft, fm);
        for (int i = 0; i < 9; i++) Rd[i] = fm[i];
        return matrix_to_vector(Rd);
    }
    Pose get_pose() const { return pose; }
    Vec6f get_vector() const { return {pose.x, pose.y, pose.z, pose.rx, pose.ry, pose.rz}; }

private:
    static void mul33(const float* a, const float* b, float* o) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                float s = 0;
                for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
                o[i * 3 + j] = s;
            }
    }
    static void rodrigues(const float r[3], double R[9]) {
        double rx = r[0], ry = r[1], rz = r[2];
        const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        if (theta < 2.220446049250313e-16) return;
        double s, c;
        svo_sincos(theta, &s, &c);
        const double c1 = 1.0 - c, it = 1.0 / theta;
        rx *= it; ry *= it; rz *= it;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int k = 0; k < 9; k++) {
            double t = c * ((k % 4 == 0) ? 1.0 : 0.0);
            t = t + c1 * rrt[k];
            R[k] = t + s * r_x[k];
        }
    }
    // cv::Rodrigues (matrix -> vector) for a proper rotation, away from theta = pi
    static Vec3f matrix_to_vector(const double R[9]) {
        const double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
        const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
        double c = (R[0] + R[4] + R[8] - 1) * 0.5;
        c = c > 1. ? 1. : (c < -1. ? -1. : c);
        const double theta = std::acos(c);
        if (s < 1e-5) return {0.f, 0.f, 0.f};
        const double vth = 1 / (2 * s) * theta;
        return {(float)(rx * vth), (float)(ry * vth), (float)(rz * vth)};
    }
    Pose pose{};
    Matx33f rot_mat{}, inv_rot_mat{};
    Vec3f angles{}, translation{};
};

inline std::ostream& operator<<(std::ostream& os, const PoseManager& pm) {
    const Pose p = pm.get_pose();
    return os << p.x << "," << p.y << "," << p.z << "," << p.rx << "," << p.ry << "," << p.rz;
}

// ---- images
struct Image8 {            // CV_8U single channel view in host memory
    const uint8_t* data;
    int cols, rows, step;
};
#ifdef SVO_FACADE_HAVE_OPENCV
inline Image8 view_of(const cv::Mat& m) {
    CV_Assert(m.type() == CV_8U);
    return Image8{m.data, m.cols, m.rows, (int)m.step};
}
#endif

// one handle (HIP stream + workspaces) per facade user
class Handle {
public:
    explicit Handle(int device = 0, int max_keypoints = 8192) { check(svo_handle_create(device, max_keypoints, &h)); }
    ~Handle() { if (h) svo_handle_destroy(h); }
    Handle(const Handle&) = delete;
    Handle& operator=(const Handle&) = delete;
    svo_handle* get() const { return h; }
private:
    svo_handle* h = nullptr;
};

// owning device buffer
template <typename T>
class DeviceArray {
public:
    DeviceArray() = default;
    explicit DeviceArray(size_t n) { resize(n); }
    void resize(size_t n) {
        void* p = nullptr;
        check(svo_device_malloc(sizeof(T) * (n ? n : 1), &p));
        mem = std::shared_ptr<void>(p, [](void* q) { svo_device_free(q); });
        count = n;
    }
    T* data() const { return static_cast<T*>(mem.get()); }
    size_t size() const { return count; }
    void upload(const Handle& h, const T* src, size_t n) { check(svo_copy_to_device(h.get(), data(), src, sizeof(T) * n)); }
    void download(const Handle& h, T* dst, size_t n) const { check(svo_copy_to_host(h.get(), dst, data(), sizeof(T) * n)); }
private:
    std::shared_ptr<void> mem;
    size_t count = 0;
};

struct DeviceImage {       // a level of a pyramid in device memory (cv::Mat's place in StereoImage)
    svo_image view{};
    std::shared_ptr<void> mem;
    int cols() const { return view.width; }
    int rows() const { return view.height; }
};

// StereoImage (stereo_slam_types.hpp:41-45): left pyramid (halfSample), right level 0, the
// Gaussian pyramid of cv::buildOpticalFlowPyramid (images only; derivatives are never stored)
struct StereoImage {
    std::vector<DeviceImage> left, right, opt_flow;
};

inline DeviceImage alloc_image(int w, int h) {
    DeviceImage im;
    void* p = nullptr;
    check(svo_device_malloc((size_t)w * h, &p));
    im.mem = std::shared_ptr<void>(p, [](void* q) { svo_device_free(q); });
    im.view = svo_image{static_cast<const uint8_t*>(p), w, h, w};
    return im;
}

// what StereoSlam::new_image does with its inputs (src/lib/stereo_slam.cpp:135-139): pyramids on the device
inline StereoImage make_stereo_image(const Handle& h, const Image8& left, const Image8& right,
                                     const CameraSettings& cam) {
    StereoImage s;
    const int L = cam.max_pyramid_levels;
    std::vector<svo_image> lv(L);
    int w = left.cols, hh = left.rows;
    for (int l = 0; l < L; l++) {
        s.left.push_back(alloc_image(w > 0 ? w : 1, hh > 0 ? hh : 1));
        s.left[l].view.width = w; s.left[l].view.height = hh; s.left[l].view.stride = w;
        lv[l] = s.left[l].view;
        w /= 2; hh /= 2;
    }
    check(svo_copy_image_to_device(h.get(), const_cast<uint8_t*>(s.left[0].view.data), left.cols, left.data,
                                   left.step, left.cols, left.rows));
    check(svo_build_pyramid(h.get(), L, lv.data()));
    s.right.push_back(alloc_image(right.cols, right.rows));
    check(svo_copy_image_to_device(h.get(), const_cast<uint8_t*>(s.right[0].view.data), right.cols, right.data,
                                   right.step, right.cols, right.rows));
    std::vector<svo_image> lk(SVO_LK_LEVELS);
    s.opt_flow.push_back(s.left[0]);                       // level 0 is the image itself
    lk[0] = s.left[0].view;
    w = left.cols; hh = left.rows;
    for (int l = 1; l < SVO_LK_LEVELS; l++) {
        w = (w + 1) / 2; hh = (hh + 1) / 2;
        s.opt_flow.push_back(alloc_image(w, hh));
        lk[l] = s.opt_flow[l].view;
    }
    int n_lk = 0;
    check(svo_build_lk_pyramid(h.get(), SVO_LK_LEVELS, cam.window_size_opt_flow, lk.data(), &n_lk));
    s.opt_flow.resize(n_lk);
    check(svo_handle_synchronize(h.get()));
    return s;
}

// ---- keypoints, frames
struct KeyPoints {
    std::vector<KeyPoint2d> kps2d;
    std::vector<KeyPoint3d> kps3d;
    std::vector<KeyPointInformation> info;
};

struct Frame {
    uint64_t id = 0;
    PoseManager pose;
    StereoImage stereo_image;
    KeyPoints kps;
    double time_stamp = 0;
};
struct KeyFrame : Frame {};

inline uint32_t flags_of(const KeyPointInformation& i) {
    return (i.ignore_during_refinement ? SVO_IGNORE_DURING_REFINEMENT : 0u) |
           (i.ignore_completely ? SVO_IGNORE_COMPLETELY : 0u) | (i.ignore_temporary ? SVO_IGNORE_TEMPORARY : 0u);
}
inline void set_flags(KeyPointInformation& i, uint32_t f) {
    i.ignore_during_refinement = (f & SVO_IGNORE_DURING_REFINEMENT) != 0;
    i.ignore_completely = (f & SVO_IGNORE_COMPLETELY) != 0;
    i.ignore_temporary = (f & SVO_IGNORE_TEMPORARY) != 0;
}

// KeyFrameManager (src/include/keyframe_manager.hpp): the container side only — keyframes are
// created inside the tracker (StereoSlam); the stage classes look them up by id
class KeyFrameManager {
public:
    explicit KeyFrameManager(const CameraSettings& camera_settings) : camera_settings(camera_settings) {}
    KeyFrame* add_keyframe(const KeyFrame& kf) { keyframes.push_back(kf); keyframes.back().id = keyframes.size() - 1; return &keyframes.back(); }
    KeyFrame* get_keyframe(uint32_t id) { return id < keyframes.size() ? &keyframes[id] : nullptr; }
    void get_keyframes(std::vector<KeyFrame>& out) const { out = keyframes; }
private:
    const CameraSettings camera_settings;
    std::vector<KeyFrame> keyframes;
};

}  // namespace svo_amd
