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

#ifdef __cplusplus
}
#endif
#endif
