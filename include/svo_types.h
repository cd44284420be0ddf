/*
 * svo_types.h — plain-C data types shared by the HIP product library
 * (include/svo_hip.h) and by the CPU oracle (oracle/svo_oracle.h).
 *
 * Every type mirrors a reference type WITHOUT OpenCV, so that a C++ facade
 * with the reference's class names can forward to the C ABI:
 *
 *   svo_camera_settings  <- CameraSettings        src/include/stereo_slam_types.hpp:16-36 (same field order)
 *   svo_kp2d / svo_kp3d  <- KeyPoint2d/KeyPoint3d src/include/stereo_slam_types.hpp:58-70
 *   svo_pose             <- Pose                  src/include/pose_manager.hpp:21-28
 *   svo_kp_info          <- KeyPointInformation   src/include/stereo_slam_types.hpp:85-98
 *                           (the embedded cv::KalmanFilter is replaced by the
 *                            two floats it really carries: state 1/z and its variance)
 *   svo_keypoints        <- KeyPoints             src/include/stereo_slam_types.hpp:106-110 (SoA of vectors)
 *   svo_image            <- cv::Mat (CV_8U, 1 channel) as used by StereoImage, :41-45
 */
#ifndef SVO_TYPES_H
#define SVO_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svo_camera_settings {
    float baseline; /* baseline [m] * fx [px]  => z = baseline / disparity */
    float fx, fy, cx, cy;
    float k1, k2, k3, p1, p2;
    int32_t grid_height;
    int32_t grid_width;
    int32_t search_x;
    int32_t search_y;
    int32_t window_size_pose_estimator;
    int32_t window_size_opt_flow;
    int32_t window_size_depth_calculator;
    int32_t max_pyramid_levels;
    int32_t min_pyramid_level_pose_estimation;
} svo_camera_settings;

typedef struct svo_kp2d { float x, y; } svo_kp2d;
typedef struct svo_kp3d { float x, y, z; } svo_kp3d;
typedef struct svo_pose { float x, y, z, rx, ry, rz; } svo_pose;

/* 8-bit single channel image view (row-major). For the HIP library `data`
 * is a device pointer unless a function says otherwise. */
typedef struct svo_image {
    const uint8_t *data;
    int32_t width;
    int32_t height;
    int32_t stride; /* bytes between rows */
} svo_image;

enum { SVO_KP_FAST = 0, SVO_KP_EDGELET = 1 };

/* bits of svo_keypoints.flags[i] */
enum {
    SVO_IGNORE_DURING_REFINEMENT = 1u << 0,
    SVO_IGNORE_COMPLETELY        = 1u << 1,
    SVO_IGNORE_TEMPORARY         = 1u << 2
};

/* Host-side AoS record returned by the getters (one per keypoint). */
typedef struct svo_kp_info {
    float    score;
    int32_t  level;
    int32_t  type;            /* SVO_KP_FAST / SVO_KP_EDGELET */
    int32_t  keyframe_id;
    int32_t  keypoint_index;
    uint8_t  color[3];
    uint8_t  ignore_during_refinement;
    uint8_t  ignore_completely;
    uint8_t  ignore_temporary;
    uint8_t  _pad[2];
    int32_t  outlier_count;
    int32_t  inlier_count;
    float    kf_inv_depth;    /* cv::KalmanFilter statePost(0)   = 1/z      */
    float    kf_variance;     /* cv::KalmanFilter errorCovPost(0,0)         */
} svo_kp_info;

/* Struct-of-arrays keypoint set: what the kernels work on. All arrays have
 * `n` valid entries (capacity is owned by whoever allocated them). */
typedef struct svo_keypoints {
    int32_t   n;
    svo_kp2d *kps2d;          /* [n]                                        */
    svo_kp3d *kps3d;          /* [n]                                        */
    uint32_t *flags;          /* [n] SVO_IGNORE_* bits                      */
    int32_t  *keyframe_id;    /* [n]                                        */
    int32_t  *keypoint_index; /* [n] index inside the origin keyframe       */
    int32_t  *outlier_count;  /* [n]                                        */
    int32_t  *inlier_count;   /* [n]                                        */
    float    *kf_inv_depth;   /* [n]                                        */
    float    *kf_variance;    /* [n]                                        */
    float    *score;          /* [n]                                        */
    int32_t  *level_type;     /* [n] level | (type << 8)                    */
    uint32_t *color;          /* [n] r | g<<8 | b<<16                       */
} svo_keypoints;

/* What one Gauss-Newton run did on one pyramid level (or the one level of
 * the reprojection GN). Mirrors the control flow of
 * PoseEstimator::estimate_pose_at_level, src/lib/pose_estimator.cpp:166-222. */
typedef struct svo_gn_trace {
    int32_t level;
    int32_t n_gradient;     /* get_gradient calls                            */
    int32_t n_cost;         /* do_calc calls (incl. the initial one)         */
    int32_t n_accepted;     /* accepted steps                                */
    int32_t exit_small;     /* 1: left through |dcost| < eps, 0: iteration cap */
    float   initial_cost;
    float   final_cost;
    float   pose[6];        /* pose at the end of the level                  */
} svo_gn_trace;

#define SVO_MAX_PYRAMID_LEVELS 8
#define SVO_LK_LEVELS 3        /* maxLevel = 2, src/lib/optical_flow.cpp:42   */

#ifdef __cplusplus
}
#endif
#endif
