// svo_kernels.hpp — argument blocks and launchers of the gfx950 kernels.
//
// Every kernel takes a device array of per-sequence argument blocks and uses
// blockIdx.{z|y} as the sequence index, so B independent sequences (one
// StereoSlam instance each) share every launch. A single sequence is B = 1.
#pragma once

#include <hip/hip_runtime.h>
#include <mutex>

#include "svo_device.hpp"

namespace svo {

// A kernel's dynamic-LDS limit belongs to the device that is current when it is raised: once per
// device and kernel (a second sequence group's thread must not launch before the first has raised it).
// Returns the HIP error of the attribute call, so a failure surfaces at the launch that needs it.
constexpr int SVO_MAX_DEVICES = 64;
struct LdsLimit {
    std::once_flag once[SVO_MAX_DEVICES];
    hipError_t err[SVO_MAX_DEVICES];
};
inline hipError_t raise_lds_limit(LdsLimit& st, const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= SVO_MAX_DEVICES) return hipErrorInvalidDevice;
    std::call_once(st.once[dev], [&st, dev, kernel, bytes] {
        st.err[dev] = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    });
    return st.err[dev];
}

// ---------------------------------------------------------------- pyramids
struct PyrArgs {
    ImgView level[SVO_MAX_PYRAMID_LEVELS];  // halfSample pyramid: [0] = input (or its resident copy), [1..] = outputs
    int n_levels;
    ImgView lk[SVO_LK_LEVELS];              // Gaussian (LK) pyramid: [0] = level[0], [1..] = outputs
    int n_lk;                               // LK levels to build (<= 1: none)
    // optional ingest of device-resident caller images (svo_new_images, SVO_MEM_DEVICE):
    // level[0] is then WRITTEN from src_left while the pyramid is built, and the
    // right image is copied by the blocks with blockIdx.z >= batch.
    ImgView src_left, src_right, dst_right;
};
// both pyramids of `batch` left images of w x h in one launch; right_blocks: extra workgroups copy
// src_right -> dst_right (ingest of device-resident frames)
// stream_rows: 0 if pyr_stream_rows() is 0 for any argument block of the launch (then the tile kernel), else the
// largest of them (the row-streaming kernel with that row block)
void launch_pyr_fused(const PyrArgs* d_args, int batch, int w, int h, bool right_blocks, int stream_rows, hipStream_t stream);
int pyr_stream_rows(const PyrArgs& host_args);

// ------------------------------------------------- sparse image alignment
struct SiaArgs {
    ImgView prev[SVO_MAX_PYRAMID_LEVELS];
    ImgView cur[SVO_MAX_PYRAMID_LEVELS];
    svo_camera_settings cam;
    const int* n_ptr;             // number of keypoints (device)
    const svo_kp2d* kps2d;        // previous frame, full resolution
    const svo_kp3d* kps3d;
    const uint32_t* flags;        // SVO_IGNORE_TEMPORARY => not used
    const float* pose_guess;      // [6]
    float* pose_out;              // [6]
    float* cost_out;              // [1]
    svo_gn_trace* trace;          // [SVO_MAX_PYRAMID_LEVELS] or null
    float* rec_ws;                // workspace [levels used][68][rec_cap]: per-level records of sia_prep_kernel
    int rec_cap;                  // keypoint capacity (row length) of rec_ws, multiple of 4
    float* kp_ws;                 // workspace [9][rec_cap] floats: per-keypoint values of sets that do not fit LDS
    float* dbg_H;                 // optional [36+6+6]: H, b, step of the first get_gradient of `dbg_level`
    int dbg_level;
    int cap;
    int exact_pinv;               // 1: reference-order normal equations + the reference's SVD pseudo-inverse (parity mode)
    PoseMats* mats_out;           // optional: rotation matrices of pose_out, for the kernels that project with it
};
// n_bound: upper bound of the keypoint counts of the launch's sequences (chooses the workgroup
// shape); rec_cap: SiaArgs::rec_cap of every block; exact: the value of SiaArgs::exact_pinv in every block (sizes the LDS staging).
// false: capacity, or the LDS limit could not be raised (then hipGetLastError() reports it)
bool launch_sia(const SiaArgs* d_args, int batch, const svo_camera_settings& cam, int width,
                int height, int n_bound, int rec_cap, int exact, hipStream_t stream);   // false: capacity
size_t sia_rec_ws_floats(const svo_camera_settings& cam, int rec_cap);   // size of SiaArgs::rec_ws

// --------------------------------------------------------------------- KLT
struct KfDev {                    // one keyframe as the device sees it
    ImgView lk[SVO_LK_LEVELS];    // Gaussian pyramid (unpadded)
    int n_lk;
    float pose[6];
    svo_kp2d* kps2d;              // keyframe->kps.kps2d
    svo_kp3d* kps3d;              // keyframe->kps.kps3d (updated by the depth filter)
    uint32_t* flags;              // ignore_temporary / ignore_completely mirror
    int* outlier_count;
    int* inlier_count;
    // the rest of frame.kps.info as copied at creation (keyframe_manager.cpp:27): read by the getters only
    int* kf_id; int* kp_index; float* score; int* level_type; uint32_t* color; float* kfx; float* kfP;
    int n;
    // KLT template cache (klt.hip): [tmpl_cap][SVO_LK_LEVELS] records of klt_template_bytes(tmpl_win) and their
    // "stored" flags; null when the keyframe has none (stage API, or evicted from the sequence's ring)
    void* tmpl;
    uint8_t* tmpl_valid;
    int tmpl_cap, tmpl_win;
};

struct KltArgs {
    const KfDev* kfs;             // keyframe table
    const int* kf_id;             // [n] origin keyframe of each point (null: all 0)
    ImgView cur[SVO_LK_LEVELS];
    int n_cur;
    const int* n_ptr;
    const svo_kp2d* prev_pts;     // [n] reference positions
    svo_kp2d* cur_pts;            // [n] in: initial flow, out: tracked
    uint8_t* status;              // [n]
    float* err;                   // [n]
    int win;
    // optional fused projection (tracker path): cur_pts = project(pose, kps3d) first
    const float* proj_pose;       // [6] or null
    const PoseMats* proj_mats;    // optional: pose_mats(proj_pose) computed once per sequence (sia_gn_kernel)
    const svo_kp3d* kps3d;
    svo_kp2d* proj_out;           // [n] projected positions (frame.kps.kps2d before the merge)
    const int* kp_index;          // [n] with proj_pose: prev_pts is gathered from kfs[kf_id].kps2d[kp_index]
    svo_kp2d* ref_out;            // [n] gathered reference points (or null)
    svo_camera_settings cam;
};
void launch_klt(const KltArgs* d_args, int batch, int max_n, int win, hipStream_t stream);
size_t klt_template_bytes(int win);

// ------------------------------------------- merge + reprojection GN (B1,B3)
struct ReprojArgs {
    svo_camera_settings cam;
    const int* n_ptr;
    svo_kp2d* kps2d;              // in: projected, out: merged
    const svo_kp3d* kps3d;
    uint32_t* flags;
    const svo_kp2d* tracked;      // null: skip the merge
    const float* err;
    const float* pose_in;
    float* pose_out;
    float* cost_out;
    svo_gn_trace* trace;          // [1] or null
    int exact_pinv;
    int* zero_out;                // or null: set to 0 (the inside counter filter_update_kernel adds to)
};
bool launch_reproj(const ReprojArgs* d_args, int batch, int n_bound, hipStream_t stream);   // false: too many keypoints

// project_keypoints (src/lib/transform_keypoints.cpp:11-48) as a stage of its own (the tracker
// fuses it into klt_track_kernel / filter_update_kernel)
void launch_project(const float* pose, const svo_kp3d* kps3d, int n, const svo_camera_settings& cam,
                    svo_kp2d* out, hipStream_t stream);

// ------------------------------------------------------------ depth filter
struct SsdArgs {
    ImgView left, right;
    const int* n_ptr;
    const svo_kp2d* kps2d;
    float* disparity;
    int win, search_x, search_y, clamp_half;
    int first;                    // process kps [first, n)
    const int* first_ptr;         // optional device value added to `first`
    const int* enable;            // optional device predicate
};
// win / search_y: the largest SsdArgs::win / search_y of the launch (choose the LDS size of the kernel)
void launch_ssd(const SsdArgs* d_args, int batch, int max_n, int win, int search_y, hipStream_t stream);

struct FilterArgs {
    svo_camera_settings cam;
    const int* n_ptr;
    const float* frame_pose;      // [6] refined pose
    svo_kp2d* kps2d;              // in: merged positions; out (if reproject): project(pose, kps3d)
    svo_kp3d* kps3d;
    uint32_t* flags;
    int* outlier_count;
    int* inlier_count;
    float* kf_inv_depth;
    float* kf_variance;
    const float* disparity;
    // explicit per-point references (stage API) ...
    const svo_kp3d* ref3d;
    const svo_kp2d* ref2d;
    const float* kf_pose;         // [n*6]
    // ... or the keyframe table (tracker): gathers refs and writes results back
    KfDev* kfs;
    const int* kf_id;
    const int* kp_index;
    int do_outlier_check, do_update, do_flags, do_reproject;
    int width, height;
    int* inside_count;            // keyframe_needed numerator (or null)
};
void launch_filter(const FilterArgs* d_args, int batch, int max_n, hipStream_t stream);   // inside_count must be zero

}  // namespace svo
