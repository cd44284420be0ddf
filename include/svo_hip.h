/*
 * svo_hip.h — C ABI of libsvo_hip.so, the MI355X (gfx950) implementation of the
 * stereo-SVO hot path. No torch / OpenCV / C++ types cross this boundary.
 *
 * The reference library (libstereosvo.so) has no C ABI: its consumers link
 * mangled C++ symbols (src/app/Makefile:17-18, src/python/setup.py:29-33).
 * Each entry point below names the reference interface it replaces; the C++
 * facade in stereo-svo-slam_amd/hostcpp/ and INTEGRATION.md show the binding a
 * maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success, < 0 on error
 * (svo_last_error() has the text); nothing throws. Unless stated otherwise
 * all data pointers are DEVICE pointers and work is enqueued on the handle's
 * HIP stream without synchronising.
 */
#ifndef SVO_HIP_H
#define SVO_HIP_H

#include <stddef.h>

#include "svo_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SVO_OK = 0,
    SVO_ERR_INVALID = -1,
    SVO_ERR_HIP = -2,
    SVO_ERR_NO_DEVICE = -3,
    SVO_ERR_CAPACITY = -4
};

typedef struct svo_handle svo_handle;

const char *svo_last_error(void);
int svo_version(void);

/* one handle per (GPU, caller thread): stream + small workspaces */
int svo_handle_create(int device, int max_keypoints, svo_handle **out);
int svo_handle_destroy(svo_handle *h);
int svo_handle_set_stream(svo_handle *h, void *hip_stream); /* hipStream_t, NULL = default */
int svo_handle_synchronize(svo_handle *h);
/* The two Gauss-Newton kernels (sparse alignment, reprojection) have two forms of the normal
 * equations. Default (fast_solver = 0): the reference's arithmetic — `hessian += row^T row`,
 * `residual -= row * diff` accumulated row by row in storage order and the Jacobi-SVD
 * pseudo-inverse of Matx66f::inv(DECOMP_SVD) (src/lib/pose_estimator.cpp:399-405,472-477,
 * pose_refinement.cpp:393-398): poses, costs and iteration traces are the CPU restatement's, bit
 * for bit. fast_solver = 1: J^T (sum g g^T) J per keypoint summed in a tree and an LDL^T solve in
 * double (SVD only for rank deficient systems): ~1.5x the frame rate, same minimum within
 * 1e-4 m / rad on smooth motion, but not the reference's iteration trace.
 * (The cost is summed in the reference's order in both.) */
int svo_handle_set_fast_solver(svo_handle *h, int on);
int svo_handle_set_exact_pinv(svo_handle *h, int on);   /* older name: set_fast_solver(!on) */

/* device memory for callers without a HIP toolchain (the C++ facades in
 * stereo-svo-slam_amd/hostcpp/, ctypes): plain hipMalloc / hipFree and copies that are
 * ordered on the handle's stream and complete on return */
int svo_device_malloc(size_t bytes, void **out);
int svo_device_free(void *p);
int svo_copy_to_device(svo_handle *h, void *dst, const void *src, size_t bytes);
int svo_copy_to_host(svo_handle *h, void *dst, const void *src, size_t bytes);
/* rows of `width` bytes: host image (src_stride) -> device image (dst_stride) */
int svo_copy_image_to_device(svo_handle *h, void *dst, size_t dst_stride, const void *src,
                             size_t src_stride, size_t width, size_t height);

/* ---- stage level entry points (one per row of SURVEY §8a) ----------------
 * P1  createImgPyramid / halfSample          src/lib/stereo_slam.cpp:93-121
 * levels[0] = input; levels[1..n-1].data = caller-allocated outputs
 * (width/height/stride are filled in; stride = width). */
int svo_build_pyramid(svo_handle *h, int n_levels, svo_image *levels);
/* P2  image part of cv::buildOpticalFlowPyramid(.., Size(win,win), 2)
 *                                              src/lib/stereo_slam.cpp:137-139
 * levels[l>=1] receive pyrDown of the previous level; *n_out = usable levels. */
int svo_build_lk_pyramid(svo_handle *h, int max_levels, int win, svo_image *levels, int *n_out);

/* A   PoseEstimator::estimate_pose(guess, out) src/include/pose_estimator.hpp:19-27,
 *                                              src/lib/pose_estimator.cpp:115-130
 * prev_pyr/cur_pyr: cam->max_pyramid_levels halfSample levels (host array of
 * views onto device memory). pose_guess/pose_out/cost/trace: device memory;
 * trace = [SVO_MAX_PYRAMID_LEVELS] svo_gn_trace or NULL.
 * dbg (optional, device, 48 floats): H, b, step of the first get_gradient on dbg_level. */
int svo_sparse_align(svo_handle *h, const svo_image *prev_pyr, const svo_image *cur_pyr,
                     const svo_kp2d *kps2d, const svo_kp3d *kps3d, const uint32_t *flags, int n,
                     const svo_camera_settings *cam, const float *pose_guess, float *pose_out,
                     float *cost, svo_gn_trace *trace, float *dbg, int dbg_level);

/* A3  project_keypoints(pose, in, camera_settings, out)   src/include/transform_keypoints.hpp:17-19,
 *                                              src/lib/transform_keypoints.cpp:11-48
 * pose: 6 floats in device memory. */
int svo_project_keypoints(svo_handle *h, const float *pose, const svo_kp3d *kps3d, int n,
                          const svo_camera_settings *cam, svo_kp2d *out);

/* B2  OpticalFlow::calculate_optical_flow      src/include/optical_flow.hpp:26-30,
 *                                              src/lib/optical_flow.cpp:14-56 */
int svo_klt_track(svo_handle *h, const svo_image *prev_lk, const svo_image *cur_lk, int n_levels,
                  const svo_kp2d *prev_pts, svo_kp2d *cur_pts, int n, int win,
                  uint8_t *status, float *err);

/* B1+B3  merge of PoseRefiner::refine_pose + PoseRefiner::update_pose
 *                                              src/lib/pose_refinement.cpp:125-150,236-290
 * tracked/err may be NULL (no merge). */
int svo_reproj_gn(svo_handle *h, svo_kp2d *kps2d, const svo_kp3d *kps3d, uint32_t *flags, int n,
                  const svo_camera_settings *cam, const svo_kp2d *tracked, const float *err,
                  const float *pose_in, float *pose_out, float *cost, svo_gn_trace *trace);

/* C1  DepthFilter::calculate_disparities       src/lib/depth_filter.cpp:259-327
 *     (clamp_half = 0: the loop of DepthCalculator::calculate_depth,
 *      src/lib/depth_calculator.cpp:200-240) */
int svo_ssd_disparity(svo_handle *h, const svo_image *left, const svo_image *right,
                      const svo_kp2d *kps2d, int n, int win, int search_x, int search_y,
                      int clamp_half, float *disparity);

/* C2+D1  DepthFilter::outlier_check + update_kps3d
 *                                              src/lib/depth_filter.cpp:52-128,130-257
 * ref3d/ref2d = keyframe->kps.kps3d/kps2d[keypoint_index], kf_pose = n*6 floats. */
int svo_depth_filter_update(svo_handle *h, const svo_kp2d *kps2d, svo_kp3d *kps3d,
                            const uint32_t *flags, int n, const svo_camera_settings *cam,
                            const float *frame_pose, const float *disparity,
                            const svo_kp3d *ref3d, const svo_kp2d *ref2d, const float *kf_pose,
                            int32_t *outlier_count, int32_t *inlier_count,
                            float *kf_inv_depth, float *kf_variance,
                            int do_outlier_check, int do_update);

/* ---- whole tracker: StereoSlam (src/include/stereo_slam.hpp:27-79) --------
 * One svo_ctx owns `n_sequences` independent StereoSlam instances; the sequences of
 * a group share every kernel launch (sequence = a grid dimension);
 * n_sequences = 1 is the drop-in for one StereoSlam object. A ctx is
 * single-caller; different ctxs are independent (own stream, own counters —
 * the reference's process-global keyframe counters, keyframe_manager.cpp:8 and
 * depth_calculator.cpp:135, are per sequence here). */
typedef struct svo_ctx svo_ctx;

/* SVO_MEM_DEVICE_BORROW: device images that the ctx uses IN PLACE as level 0 of its pyramids and
 * as the right image — no copy, like the reference, whose level 0 is a shallow alias of the caller's
 * cv::Mat (src/lib/stereo_slam.cpp:115). The caller keeps every image valid and unchanged while a
 * frame or keyframe of the ctx refers to it (the safe choice: until svo_ctx_destroy). */
enum { SVO_MEM_HOST = 0, SVO_MEM_DEVICE = 1, SVO_MEM_DEVICE_BORROW = 2 };

/* StereoSlam::StereoSlam(const CameraSettings&)         src/lib/stereo_slam.cpp:29-41 */
int svo_ctx_create(const svo_camera_settings *cam, int width, int height, int n_sequences,
                   int device, svo_ctx **out);
int svo_ctx_destroy(svo_ctx *ctx);

/* StereoSlam::new_image(left, right, time_stamp)        src/lib/stereo_slam.cpp:123-271
 * for every sequence of the ctx: left[s]/right[s] point to 8-bit images of the
 * ctx size with `stride` bytes per row, in host (SVO_MEM_HOST) or device memory.
 * The images are copied; the caller may reuse its buffers on return. Returns
 * after the frame is complete (like the reference). A sequence whose two pointers are
 * NULL sits the step out with its state untouched (sequences of one ctx may have different
 * lengths); the very first step needs every sequence. */
int svo_new_images(svo_ctx *ctx, const uint8_t *const *left, const uint8_t *const *right,
                   int stride, const float *time_stamps, int mem);
/* Pipelined form (no counterpart in the reference, whose new_image is synchronous): svo_submit_images() queues one frame set (same arguments; the images,
 * host or device, must stay valid until svo_wait) on every sequence group and returns; svo_wait()
 * blocks until all queued frame sets are processed and reports the first error. The groups
 * (svo_ctx_get_groups; SVO_GROUPS overrides the default) advance independently, each on its
 * own HIP stream and host thread, so one group's host round trips (keyframe decision,
 * argument blocks) overlap the other groups' kernels. svo_new_images = submit + wait; every
 * getter waits first. */
int svo_submit_images(svo_ctx *ctx, const uint8_t *const *left, const uint8_t *const *right,
                      int stride, const float *time_stamps, int mem);
int svo_wait(svo_ctx *ctx);
int svo_ctx_get_groups(svo_ctx *ctx, int *n_groups);
/* n_sequences == 1, host memory: the exact shape of StereoSlam::new_image */
int svo_new_image(svo_ctx *ctx, const uint8_t *left, int left_stride, const uint8_t *right,
                  int right_stride, int width, int height, float time_stamp);

/* Frame::pose of the current frame (get_frame, src/lib/stereo_slam.cpp:278-284) */
int svo_get_pose(svo_ctx *ctx, int seq, float pose[6]);
/* Frame::kps of the current frame; returns the count in *n (copies min(n, cap)) */
int svo_get_frame_keypoints(svo_ctx *ctx, int seq, svo_kp2d *kps2d, svo_kp3d *kps3d,
                            svo_kp_info *info, int cap, int *n);
/* get_keyframes / get_keyframe (src/lib/stereo_slam.cpp:273-289) */
int svo_get_keyframe_count(svo_ctx *ctx, int seq, int *count);
int svo_get_keyframe(svo_ctx *ctx, int seq, int id, svo_kp2d *kps2d, svo_kp3d *kps3d,
                     svo_kp_info *info, float pose[6], int cap, int *n);
/* get_trajectory (src/lib/stereo_slam.cpp:291-294) */
int svo_get_trajectory(svo_ctx *ctx, int seq, svo_pose *out, int cap, int *n);
/* StereoSlam::update_pose (12-state Kalman)             src/lib/stereo_slam.cpp:296-359 */
int svo_update_pose(svo_ctx *ctx, int seq, const float pose[6], const float speed[6],
                    const float pose_var[6], const float speed_var[6], double dt,
                    float filtered[6]);

/* per-frame diagnostics of the last svo_new_images call */
typedef struct svo_frame_stats {
    int32_t frame_id;
    int32_t is_keyframe;
    int32_t n_keypoints;
    int32_t n_keyframes;
    int32_t inside_count;
    int32_t overflow;
    float   pose_sia[6];
    float   pose_refined[6];
    float   sia_cost, reproj_cost;
    float   sia_ms;             /* device time of the sparse-alignment kernel (timing on) */
    float   stage_ms[8];        /* HIP-event times on the ctx stream (timing on): images+pyramids,
                                   compaction, sparse alignment, KLT, merge+reprojection GN,
                                   SSD disparity, filter update, keyframe phase + read-back */
    svo_gn_trace sia_trace[SVO_MAX_PYRAMID_LEVELS];
    svo_gn_trace reproj_trace;
} svo_frame_stats;
int svo_get_frame_stats(svo_ctx *ctx, int seq, svo_frame_stats *out);
/* counters accumulated over all sequences and all svo_new_images calls so far */
typedef struct svo_totals {
    int64_t frames;             /* sequence-frames processed                      */
    int64_t keyframes;          /* of which created a keyframe                    */
    int64_t keypoints;          /* sum of n_keypoints                             */
    int64_t gn_gradient_calls;  /* sum of get_gradient calls of the sparse alignment */
    int64_t gn_cost_calls;
    double  stage_ms[8];        /* sum of svo_frame_stats.stage_ms (timing on)    */
    double  wall_ms;            /* host wall time spent inside svo_new_images (max over groups) */
    int64_t launches;           /* frame sets processed, summed over groups: stage_ms / launches =
                                   mean duration of one stage launch                */
    int32_t n_groups;           /* independently driven sequence groups of the ctx */
    int32_t image_sets;         /* image sets (pyramids of one frame) allocated so far, summed over groups: bounded by
                                   the keyframes whose keypoints are still tracked (their images are released after that) */
} svo_totals;
int svo_get_totals(svo_ctx *ctx, svo_totals *out);
int svo_ctx_enable_timing(svo_ctx *ctx, int on);
int svo_ctx_set_fast_solver(svo_ctx *ctx, int on);  /* see svo_handle_set_fast_solver; default 0 */
int svo_ctx_set_exact_pinv(svo_ctx *ctx, int on);   /* older name: set_fast_solver(!on) */

#ifdef __cplusplus
}
#endif
#endif
