/*
 * svo_oracle.h — CPU ORACLE for the stereo-SVO hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is a plain-C restatement of the reference's algorithm
 * (eichenberger/stereo-svo-slam, the .cpp files under src/lib) for the path
 *   pyramids -> sparse image alignment (pose GN) -> KLT refinement ->
 *   reprojection GN -> stereo depth filter.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product (stereo-svo-slam_amd/csrc) never links or calls it.
 *
 * PARITY STATUS
 *  - pinned by the reference's own fixtures: exponential map known-answer
 *    (src/test/test_exponential_map.cpp:35-48) and the real stereo pair
 *    src/test/left.png / right.png used as inputs (tests/golden/).
 *  - PARITY UNPINNED for everything whose arithmetic lives in OpenCV 4.x
 *    (un-vendored, version unpinned, not installed here): cv::Rodrigues,
 *    cv::projectPoints, Matx::inv(DECOMP_SVD), cv::solve(DECOMP_SVD),
 *    cv::buildOpticalFlowPyramid, cv::calcOpticalFlowPyrLK, cv::matchTemplate,
 *    cv::KalmanFilter, cv::FAST, cv::Sobel.  These are restated from OpenCV's
 *    published algorithms as remembered; where OpenCV's own rounding cannot
 *    be known offline this file DEFINES it (see each function).
 *  - The reference itself cannot be built here (every TU includes
 *    <opencv2/opencv.hpp>), so there is no oracle/_ref.
 *
 * All pointers are HOST pointers. All functions are single threaded and
 * deterministic (build with -ffp-contract=off, no fast-math).
 */
#ifndef SVO_ORACLE_H
#define SVO_ORACLE_H

#include "../include/svo_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- restated OpenCV primitives (cv_prims.c) ---------------------------- */
void svo_o_rodrigues(const float r[3], double R[9]);
void svo_o_rodrigues_f(const float r[3], float R[9]);
/* PoseManager::set_pose, src/lib/pose_manager.cpp:9-17: R(r) and R(-r) as float */
void svo_o_pose_matrices(const float pose[6], float rot[9], float inv_rot[9]);
/* one-sided Jacobi SVD of the n x m matrix At (rows = columns of A) */
void svo_o_jacobi_svd(float *At, int astep, float *W, float *Vt, int vstep,
                      int m, int n, int n1);
/* Matx<float,n,n>::inv(DECOMP_SVD); returns 0 and writes zeros when OpenCV would */
int  svo_o_inv_svd(const float *A, int n, float *Ainv);
/* cv::solve(A[m x n], b[m], x[n], DECOMP_SVD), single right-hand side */
void svo_o_solve_svd(const float *A, int m, int n, const float *b, float *x);
void svo_o_exponential_map(const float twist[6], float out[6]);
void svo_o_project_keypoints(const float pose[6], const svo_kp3d *in, int n,
                             const svo_camera_settings *cam, svo_kp2d *out);
/* cv::pyrDown (8U, 5x5 Gaussian, BORDER_REFLECT_101); dst is ((w+1)/2)x((h+1)/2) */
void svo_o_pyr_down(const uint8_t *src, int w, int h, int sstride,
                    uint8_t *dst, int dstride);
/* calcSharrDeriv of OpenCV's lkpyramid.cpp: interleaved (dx,dy) int16 */
void svo_o_scharr(const uint8_t *src, int w, int h, int sstride, int16_t *dst /* w*h*2 */);
/* 1-state cv::KalmanFilter predict()+correct(1/z) as used by the depth filter */
void svo_o_kf1_update(float *x, float *P, float Q, float R, float meas);

/* ---- hot path (hot_path.c) ---------------------------------------------- */
/* P1: halfSample/createImgPyramid, src/lib/stereo_slam.cpp:93-121.
 * levels[0] is the input; levels[1..n-1].data must point to caller-owned
 * buffers of (w>>l)*(h>>l) bytes (stride = width). Fills width/height/stride. */
void svo_o_build_pyramid(const svo_image *lvl0, int n_levels, svo_image *levels);
/* P2: Gaussian part of cv::buildOpticalFlowPyramid(maxLevel=2): levels[l] for
 * l>=1 get pyrDown of the previous one; returns the number of usable levels
 * (OpenCV stops when a level is not larger than the window). */
int  svo_o_build_lk_pyramid(const svo_image *lvl0, int max_levels, int win, svo_image *levels);

/* A4: get_total_intensity_diff, src/lib/image_comparison.cpp:9-120 */
float svo_o_total_intensity_diff(const svo_image *img1, const svo_image *img2,
                                 const svo_kp2d *kps1, const svo_kp2d *kps2, int n, int patch);
/* A: PoseEstimator::estimate_pose, src/lib/pose_estimator.cpp:115-130.
 * kps with SVO_IGNORE_TEMPORARY are skipped (ctor, :226-259).
 * trace may be NULL, else [max_pyramid_levels] entries (index = level). */
float svo_o_sparse_align(const svo_image *prev_pyr, const svo_image *cur_pyr,
                         const svo_kp2d *kps2d, const svo_kp3d *kps3d,
                         const uint32_t *flags, int n,
                         const svo_camera_settings *cam,
                         const float pose_guess[6], float pose_out[6],
                         svo_gn_trace *trace);
/* one get_gradient call (pose_estimator.cpp:418-539) exposed for unit parity:
 * writes H (36), b (6) and the 6-vector step */
void svo_o_sia_gradient(const svo_image *prev, const svo_image *cur, int level,
                        const svo_kp2d *kps2d, const svo_kp3d *kps3d,
                        const uint32_t *flags, int n,
                        const svo_camera_settings *cam, const float pose[6],
                        float H[36], float b[6], float step[6]);

/* B2: OpticalFlow::calculate_optical_flow -> cv::calcOpticalFlowPyrLK with
 * (win,win), maxLevel=n_levels-1, 30 its, eps 0.01, OPTFLOW_USE_INITIAL_FLOW;
 * status==0 => err=+inf (src/lib/optical_flow.cpp:14-56). */
void svo_o_klt_track(const svo_image *prev_lk, const svo_image *cur_lk, int n_levels,
                     const svo_kp2d *prev_pts, svo_kp2d *cur_pts, int n, int win,
                     uint8_t *status, float *err);
/* B1: merge rule of PoseRefiner::refine_pose, src/lib/pose_refinement.cpp:125-150 */
void svo_o_refine_merge(svo_kp2d *kps2d, uint32_t *flags, const svo_kp2d *tracked,
                        const float *err, int n);
/* B3: PoseRefiner::update_pose, src/lib/pose_refinement.cpp:236-290,321-412 */
float svo_o_reproj_gn(const svo_kp2d *kps2d, const svo_kp3d *kps3d, const uint32_t *flags,
                      int n, const svo_camera_settings *cam,
                      const float pose_in[6], float pose_out[6], svo_gn_trace *trace);

/* C1: DepthFilter::calculate_disparities, src/lib/depth_filter.cpp:259-327
 * (clamp_half=1) and the same loop in DepthCalculator::calculate_depth,
 * src/lib/depth_calculator.cpp:200-240 (clamp_half=0: no max(0.5,.), no OOB skip). */
void svo_o_ssd_disparity(const svo_image *left, const svo_image *right,
                         const svo_kp2d *kps2d, int n, int win, int search_x, int search_y,
                         int clamp_half, float *disparity);
/* C2: DepthFilter::outlier_check, src/lib/depth_filter.cpp:52-128.
 * ref3d[i]  = keyframe->kps.kps3d[keypoint_index] of kp i,
 * kf_pose[i]= pose (6 floats) of the origin keyframe of kp i. */
void svo_o_outlier_check(const svo_kp2d *kps2d, const float *disparity, int n,
                         const svo_camera_settings *cam, const float frame_pose[6],
                         const svo_kp3d *ref3d, const float *kf_pose /* n*6 */,
                         int32_t *outlier_count, int32_t *inlier_count);
/* D1: DepthFilter::update_kps3d, src/lib/depth_filter.cpp:130-257.
 * ref2d[i] = keyframe->kps.kps2d[keypoint_index]. kps3d is updated in place. */
void svo_o_update_kps3d(const svo_kp2d *kps2d, svo_kp3d *kps3d, const uint32_t *flags, int n,
                        const svo_camera_settings *cam, const float frame_pose[6],
                        const svo_kp2d *ref2d, const float *kf_pose /* n*6 */,
                        int32_t *outlier_count, float *kf_inv_depth, float *kf_variance);

/* ---- whole StereoSlam restatement (tracker.c) --------------------------- */
typedef struct svo_o_slam svo_o_slam;
svo_o_slam *svo_o_slam_create(const svo_camera_settings *cam);
void  svo_o_slam_destroy(svo_o_slam *s);
/* StereoSlam::new_image, src/lib/stereo_slam.cpp:123-271. Returns 1 if the
 * frame created a keyframe. */
int   svo_o_slam_new_image(svo_o_slam *s, const uint8_t *left, const uint8_t *right,
                           int width, int height, float time_stamp);
void  svo_o_slam_get_pose(const svo_o_slam *s, float pose[6]);
int   svo_o_slam_num_keypoints(const svo_o_slam *s);
int   svo_o_slam_num_keyframes(const svo_o_slam *s);
/* copies up to cap entries; returns n */
int   svo_o_slam_get_keypoints(const svo_o_slam *s, svo_kp2d *kps2d, svo_kp3d *kps3d,
                               svo_kp_info *info, int cap);
int   svo_o_slam_get_keyframe_keypoints(const svo_o_slam *s, int id, svo_kp2d *kps2d,
                                        svo_kp3d *kps3d, svo_kp_info *info, float pose[6], int cap);
/* per-frame diagnostics of the last new_image */
typedef struct svo_o_frame_stats {
    int32_t n_tracked;             /* kps after remove_outliers              */
    int32_t n_active;              /* !ignore_temporary used by SIA          */
    int32_t sia_gradient_calls;
    int32_t sia_cost_calls;
    double  t_total, t_pyramid, t_sia, t_klt, t_reproj, t_disparity, t_filter, t_keyframe;
    float   pose_sia[6], pose_refined[6];
    svo_gn_trace sia_trace[SVO_MAX_PYRAMID_LEVELS];
    svo_gn_trace reproj_trace;
} svo_o_frame_stats;
void  svo_o_slam_get_stats(const svo_o_slam *s, svo_o_frame_stats *out);
/* StereoSlam::update_pose (12-state Kalman), src/lib/stereo_slam.cpp:296-359 */
void  svo_o_slam_update_pose(svo_o_slam *s, const float pose[6], const float speed[6],
                             const float pose_var[6], const float speed_var[6], double dt,
                             float filtered[6]);

/* keyframe creation pieces, exposed for unit parity */
/* cv::FAST(threshold, nonmax=true, TYPE_9_16): writes score image (0 where no
 * corner survives NMS) */
void svo_o_fast_score_nms(const uint8_t *img, int w, int h, int stride, int threshold,
                          uint8_t *score /* w*h */);
/* cv::Sobel(src, dst, CV_8U, 1, 0) ksize 3, saturating */
void svo_o_sobel_x_u8(const uint8_t *img, int w, int h, int stride, uint8_t *dst /* w*h */);
/* CornerDetector::detect_keypoints, src/lib/corner_detector.cpp:13-79; returns count */
int  svo_o_detect_keypoints(const uint8_t *img, int w, int h, int stride,
                            int grid_w, int grid_h, int level,
                            svo_kp2d *kps, float *score, int32_t *type, int cap);

#ifdef __cplusplus
}
#endif
#endif
