// svo_tracker.hpp — device-side data layout of one tracked sequence and the
// argument blocks of the bookkeeping / keyframe kernels (keyframe.hip).
#pragma once

#include "svo_kernels.hpp"

namespace svo {

// Struct-of-arrays keypoint set in HBM (KeyPoints, src/include/stereo_slam_types.hpp:106-110;
// KeyPointInformation :85-98 split by field so that every kernel streams only
// the arrays it touches).
struct KpsDev {
    svo_kp2d* kps2d;
    svo_kp3d* kps3d;
    uint32_t* flags;
    int* kf_id;
    int* kp_index;
    int* outl;
    int* inl;
    float* kfx;        // depth filter state 1/z
    float* kfP;        // depth filter variance
    float* score;
    int* level_type;   // level | type << 8
    uint32_t* color;
    int* n;
};

struct CompactArgs {
    KpsDev src, dst;
    int mode;          // 0: remove_outliers, 1: find_bad_keypoints
    int width, height;
    const int* enable; // optional device predicate
    int* zero;         // optional: zero_count ints set to 0 (the detection counters of the keyframe that follows)
    int zero_count;
    int* min_kf;       // optional: [3] smallest origin-keyframe id among the keypoints kept (INT_MAX: none kept), then a 64-bit
                       // mask (low word first): bit j set = some kept keypoint comes from keyframe min + j (younger ones: not reported)
};
void launch_compact(const CompactArgs* d_args, int batch, int cap, hipStream_t stream);

struct DetCell {
    float x, y, score;
    int type;
};

struct DetectArgs {
    ImgView level[SVO_MAX_PYRAMID_LEVELS];
    int n_levels;      // left.size()/2
    int grid_w, grid_h;
    DetCell* out;      // [n_levels][max_cells]
    int* n_out;        // [n_levels]
    int max_cells;
    const int* enable;
};
// grid_w x grid_h: the level-0 cell of the launch's sequences (chooses the kernel shape)
void launch_detect(const DetectArgs* d_args, int batch, int max_cells, int n_levels, int grid_w, int grid_h, hipStream_t stream);

struct MergeArgs {
    svo_camera_settings cam;
    int width, height;
    const DetCell* det;
    const int* n_det;
    int n_levels, max_cells;
    KpsDev kps;
    int cap;
    DetCell* sel;      // [max_cells]
    int* sel_level;    // [max_cells]
    int* sel_cell;     // [max_cells]
    int* occupied;     // [merge cells]
    int* old_count;
    int* overflow;
    const int* enable;
};
void launch_select_merge(const MergeArgs* d_args, int batch, int max_cells, hipStream_t stream);

struct KfInitArgs {
    svo_camera_settings cam;
    KpsDev kps;
    const int* old_count;
    const float* disparity;
    const float* frame_pose;
    int first_frame;
    int new_kf_id;
    KfDev* kfs;
    uint32_t* color_lcg;
    int* n_out;
    const int* enable;
    // the record of the new keyframe (written to kfs[new_kf_id] by the kernel: no copy of its own per keyframe),
    // the "stored" flags of its template cache block (cleared here) and the keyframe that loses the block (-1: none)
    KfDev record;
    int tmpl_valid_bytes;
    int evict_id;
};
void launch_kf_init(const KfInitArgs* d_args, int batch, hipStream_t stream);

}  // namespace svo
