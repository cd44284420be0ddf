// svo_capi.hip — the C ABI of include/svo_hip.h: handle management and the
// stage-level entry points. Each entry fills the per-sequence argument block
// of one kernel (batch = 1), stages it in the handle's device ring and
// launches on the handle's stream. There is NO CPU fallback: without a HIP
// device every call fails with SVO_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/svo_hip.h"
#include "svo_kernels.hpp"

using namespace svo;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// shared with svo_ctx.hip
int svo_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(SVO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

struct svo_handle {
    int device;
    hipStream_t stream;
    int max_kps;
    uint8_t* ring;       // device ring for argument blocks
    size_t ring_cap, ring_off;
    float* sia_kpws;     // [9][rec_cap] per-keypoint values of the BIG alignment path
    float* sia_rec;      // [7][68][rec_cap] per-level alignment records
    int rec_cap;
    KfDev* kf_one;       // 1-entry keyframe table for svo_klt_track
    int exact_pinv;
};

extern "C" const char* svo_last_error(void) { return g_err; }
extern "C" int svo_version(void) { return 100; }

extern "C" int svo_handle_create(int device, int max_keypoints, svo_handle** out) {
    if (!out || max_keypoints <= 0) return fail(SVO_ERR_INVALID, "svo_handle_create: bad arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(SVO_ERR_NO_DEVICE, "no HIP device visible: libsvo_hip has no CPU fallback");
    if (device < 0 || device >= count) return fail(SVO_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    svo_handle* h = new (std::nothrow) svo_handle();
    if (!h) return fail(SVO_ERR_INVALID, "out of host memory");
    h->device = device;
    h->stream = nullptr;
    h->max_kps = max_keypoints;
    h->exact_pinv = 1;     // reference-order Gauss-Newton by default
    h->ring_cap = 1 << 20;
    h->ring_off = 0;
    HIP_TRY(hipMalloc(&h->ring, h->ring_cap));
    HIP_TRY(hipMalloc(&h->kf_one, sizeof(KfDev)));
    h->rec_cap = (max_keypoints + 511) / 512 * 512;
    HIP_TRY(hipMalloc(&h->sia_rec, sizeof(float) * 7 * 68 * (size_t)h->rec_cap));
    HIP_TRY(hipMalloc(&h->sia_kpws, sizeof(float) * 9 * (size_t)h->rec_cap));
    *out = h;
    return SVO_OK;
}

extern "C" int svo_handle_destroy(svo_handle* h) {
    if (!h) return SVO_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->ring);
    (void)hipFree(h->sia_kpws);
    (void)hipFree(h->sia_rec);
    (void)hipFree(h->kf_one);
    delete h;
    return SVO_OK;
}

extern "C" int svo_handle_set_stream(svo_handle* h, void* s) {
    if (!h) return fail(SVO_ERR_INVALID, "null handle");
    h->stream = reinterpret_cast<hipStream_t>(s);
    return SVO_OK;
}

extern "C" int svo_handle_set_fast_solver(svo_handle* h, int on) {
    if (!h) return fail(SVO_ERR_INVALID, "null handle");
    h->exact_pinv = on == 0;
    return SVO_OK;
}

extern "C" int svo_handle_set_exact_pinv(svo_handle* h, int on) {
    if (!h) return fail(SVO_ERR_INVALID, "null handle");
    h->exact_pinv = on != 0;
    return SVO_OK;
}

extern "C" int svo_handle_synchronize(svo_handle* h) {
    if (!h) return fail(SVO_ERR_INVALID, "null handle");
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SVO_OK;
}

// copy a host block into the device ring (stream ordered); returns device address
template <typename T>
static int stage(svo_handle* h, const T& host, T** dev) {
    const size_t bytes = (sizeof(T) + 255) & ~(size_t)255;
    if (h->ring_off + bytes > h->ring_cap) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->ring_off = 0;
    }
    T* d = reinterpret_cast<T*>(h->ring + h->ring_off);
    h->ring_off += bytes;
    HIP_TRY(hipMemcpyAsync(d, &host, sizeof(T), hipMemcpyHostToDevice, h->stream));
    *dev = d;
    return SVO_OK;
}

static int stage_n(svo_handle* h, int n, int** d_n) { return stage<int>(h, n, d_n); }

#define CHECK_H(h)                                                   \
    do {                                                             \
        if (!(h)) return fail(SVO_ERR_INVALID, "null handle");       \
        HIP_TRY(hipSetDevice((h)->device));                          \
    } while (0)

extern "C" int svo_device_malloc(size_t bytes, void** out) {
    if (!out) return fail(SVO_ERR_INVALID, "svo_device_malloc: null out");
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return SVO_OK;
}
extern "C" int svo_device_free(void* p) {
    if (p) HIP_TRY(hipFree(p));
    return SVO_OK;
}
extern "C" int svo_copy_to_device(svo_handle* h, void* dst, const void* src, size_t bytes) {
    CHECK_H(h);
    if (bytes == 0) return SVO_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SVO_OK;
}
extern "C" int svo_copy_to_host(svo_handle* h, void* dst, const void* src, size_t bytes) {
    CHECK_H(h);
    if (bytes == 0) return SVO_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SVO_OK;
}
extern "C" int svo_copy_image_to_device(svo_handle* h, void* dst, size_t dst_stride, const void* src,
                                        size_t src_stride, size_t width, size_t height) {
    CHECK_H(h);
    HIP_TRY(hipMemcpy2DAsync(dst, dst_stride, src, src_stride, width, height, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SVO_OK;
}

extern "C" int svo_project_keypoints(svo_handle* h, const float* pose, const svo_kp3d* kps3d, int n,
                                     const svo_camera_settings* cam, svo_kp2d* out) {
    CHECK_H(h);
    if (!pose || !cam || n < 0 || (n > 0 && (!kps3d || !out)))
        return fail(SVO_ERR_INVALID, "svo_project_keypoints: bad arguments");
    if (n > 0) launch_project(pose, kps3d, n, *cam, out, h->stream);
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_build_pyramid(svo_handle* h, int n_levels, svo_image* levels) {
    CHECK_H(h);
    if (!levels || n_levels < 1 || n_levels > 7)
        return fail(SVO_ERR_INVALID, "svo_build_pyramid: n_levels must be 1..7");
    PyrArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.n_levels = n_levels;
    for (int l = 1; l < n_levels; l++) {
        levels[l].width = levels[l - 1].width / 2;
        levels[l].height = levels[l - 1].height / 2;
        levels[l].stride = levels[l].width;
    }
    for (int l = 0; l < n_levels; l++) pa.level[l] = make_view(levels[l]);
    PyrArgs* d;
    int rc = stage(h, pa, &d);
    if (rc) return rc;
    if (n_levels > 1) launch_pyr_fused(d, 1, levels[0].width, levels[0].height, false, pyr_stream_rows(pa), h->stream);
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_build_lk_pyramid(svo_handle* h, int max_levels, int win, svo_image* levels,
                                    int* n_out) {
    CHECK_H(h);
    if (!levels || max_levels < 1 || max_levels > SVO_LK_LEVELS)
        return fail(SVO_ERR_INVALID, "svo_build_lk_pyramid: max_levels must be 1..%d", SVO_LK_LEVELS);
    // cv::buildOpticalFlowPyramid stops when the next level is not larger than the window
    int n = max_levels, w = levels[0].width, hgt = levels[0].height;
    for (int l = 0; l < max_levels; l++) {
        if (l) { levels[l].width = w; levels[l].height = hgt; levels[l].stride = w; }
        w = (w + 1) / 2; hgt = (hgt + 1) / 2;
        if (w <= win || hgt <= win) { n = l + 1; break; }
    }
    PyrArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.n_levels = 1;
    pa.level[0] = make_view(levels[0]);
    pa.n_lk = n;
    for (int l = 0; l < n; l++) pa.lk[l] = make_view(levels[l]);
    PyrArgs* d;
    int rc = stage(h, pa, &d);
    if (rc) return rc;
    if (n > 1) launch_pyr_fused(d, 1, levels[0].width, levels[0].height, false, pyr_stream_rows(pa), h->stream);
    HIP_TRY(hipGetLastError());
    if (n_out) *n_out = n;
    return SVO_OK;
}

extern "C" int svo_sparse_align(svo_handle* h, const svo_image* prev_pyr, const svo_image* cur_pyr,
                                const svo_kp2d* kps2d, const svo_kp3d* kps3d, const uint32_t* flags,
                                int n, const svo_camera_settings* cam, const float* pose_guess,
                                float* pose_out, float* cost, svo_gn_trace* trace, float* dbg,
                                int dbg_level) {
    CHECK_H(h);
    if (!prev_pyr || !cur_pyr || !cam || !pose_guess || !pose_out || n < 0)
        return fail(SVO_ERR_INVALID, "svo_sparse_align: bad arguments");
    if (n > h->max_kps) return fail(SVO_ERR_CAPACITY, "n=%d exceeds handle capacity %d", n, h->max_kps);
    if (cam->max_pyramid_levels < 1 || cam->max_pyramid_levels > 7 ||
        cam->min_pyramid_level_pose_estimation < 0)
        return fail(SVO_ERR_INVALID, "svo_sparse_align: max_pyramid_levels must be 1..7");
    if (cam->window_size_pose_estimator != 4)   // PATCH_SIZE, src/lib/pose_estimator.cpp:68
        return fail(SVO_ERR_INVALID, "svo_sparse_align: window_size_pose_estimator must be 4");
    SiaArgs sa;
    memset(&sa, 0, sizeof(sa));
    for (int l = 0; l < cam->max_pyramid_levels; l++) {
        sa.prev[l] = make_view(prev_pyr[l]);
        sa.cur[l] = make_view(cur_pyr[l]);
    }
    sa.cam = *cam;
    int* d_n;
    int rc = stage_n(h, n, &d_n);
    if (rc) return rc;
    sa.n_ptr = d_n;
    sa.kps2d = kps2d; sa.kps3d = kps3d; sa.flags = flags;
    sa.pose_guess = pose_guess; sa.pose_out = pose_out; sa.cost_out = cost; sa.trace = trace;
    sa.kp_ws = h->sia_kpws;
    sa.rec_ws = h->sia_rec; sa.rec_cap = h->rec_cap;
    sa.dbg_H = dbg; sa.dbg_level = dbg_level;
    sa.cap = h->max_kps;
    sa.exact_pinv = h->exact_pinv;
    SiaArgs* d;
    rc = stage(h, sa, &d);
    if (rc) return rc;
    if (!launch_sia(d, 1, *cam, cur_pyr[0].width, cur_pyr[0].height, n, h->rec_cap, h->exact_pinv, h->stream)) {
        HIP_TRY(hipGetLastError());          // (the LDS limit of the kernel could not be raised on this device)
        return fail(SVO_ERR_CAPACITY, "svo_sparse_align: %d keypoints exceed the workspaces", n);
    }
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_klt_track(svo_handle* h, const svo_image* prev_lk, const svo_image* cur_lk,
                             int n_levels, const svo_kp2d* prev_pts, svo_kp2d* cur_pts, int n, int win,
                             uint8_t* status, float* err) {
    CHECK_H(h);
    if (!prev_lk || !cur_lk || n_levels < 1 || n_levels > SVO_LK_LEVELS || n < 0)
        return fail(SVO_ERR_INVALID, "svo_klt_track: bad arguments");
    if (win < 3 || win > 35) return fail(SVO_ERR_INVALID, "svo_klt_track: window must be 3..35");
    KfDev kf;
    memset(&kf, 0, sizeof(kf));
    kf.n_lk = n_levels;
    for (int l = 0; l < n_levels; l++) kf.lk[l] = make_view(prev_lk[l]);
    KfDev* d_kf;
    int rc = stage(h, kf, &d_kf);
    if (rc) return rc;
    KltArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.kfs = d_kf;
    ka.n_cur = n_levels;
    for (int l = 0; l < n_levels; l++) ka.cur[l] = make_view(cur_lk[l]);
    int* d_n;
    rc = stage_n(h, n, &d_n);
    if (rc) return rc;
    ka.n_ptr = d_n;
    ka.prev_pts = prev_pts; ka.cur_pts = cur_pts; ka.status = status; ka.err = err; ka.win = win;
    KltArgs* d;
    rc = stage(h, ka, &d);
    if (rc) return rc;
    launch_klt(d, 1, n, win, h->stream);
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_reproj_gn(svo_handle* h, svo_kp2d* kps2d, const svo_kp3d* kps3d, uint32_t* flags,
                             int n, const svo_camera_settings* cam, const svo_kp2d* tracked,
                             const float* err, const float* pose_in, float* pose_out, float* cost,
                             svo_gn_trace* trace) {
    CHECK_H(h);
    if (!cam || !pose_in || !pose_out || n < 0) return fail(SVO_ERR_INVALID, "svo_reproj_gn: bad arguments");
    ReprojArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.cam = *cam;
    int* d_n;
    int rc = stage_n(h, n, &d_n);
    if (rc) return rc;
    ra.n_ptr = d_n;
    ra.kps2d = kps2d; ra.kps3d = kps3d; ra.flags = flags; ra.tracked = tracked; ra.err = err;
    ra.pose_in = pose_in; ra.pose_out = pose_out; ra.cost_out = cost; ra.trace = trace;
    ra.exact_pinv = h->exact_pinv;
    ReprojArgs* d;
    rc = stage(h, ra, &d);
    if (rc) return rc;
    if (!launch_reproj(d, 1, n, h->stream)) {
        HIP_TRY(hipGetLastError());
        return fail(SVO_ERR_CAPACITY, "svo_reproj_gn: %d keypoints do not fit LDS", n);
    }
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_ssd_disparity(svo_handle* h, const svo_image* left, const svo_image* right,
                                 const svo_kp2d* kps2d, int n, int win, int search_x, int search_y,
                                 int clamp_half, float* disparity) {
    CHECK_H(h);
    if (!left || !right || n < 0) return fail(SVO_ERR_INVALID, "svo_ssd_disparity: bad arguments");
    if (win < 1 || win > 35 || search_x < 0 || search_x > 64 || search_y < 0 || search_y > 8)
        return fail(SVO_ERR_INVALID, "svo_ssd_disparity: win<=35, search_x<=64, search_y<=8 supported");
    SsdArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.left = make_view(*left); sa.right = make_view(*right);
    int* d_n;
    int rc = stage_n(h, n, &d_n);
    if (rc) return rc;
    sa.n_ptr = d_n; sa.kps2d = kps2d; sa.disparity = disparity;
    sa.win = win; sa.search_x = search_x; sa.search_y = search_y; sa.clamp_half = clamp_half;
    sa.first = 0;
    SsdArgs* d;
    rc = stage(h, sa, &d);
    if (rc) return rc;
    launch_ssd(d, 1, n, win, search_y, h->stream);
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_depth_filter_update(svo_handle* h, const svo_kp2d* kps2d, svo_kp3d* kps3d,
                                       const uint32_t* flags, int n, const svo_camera_settings* cam,
                                       const float* frame_pose, const float* disparity,
                                       const svo_kp3d* ref3d, const svo_kp2d* ref2d,
                                       const float* kf_pose, int32_t* outlier_count,
                                       int32_t* inlier_count, float* kf_inv_depth, float* kf_variance,
                                       int do_outlier_check, int do_update) {
    CHECK_H(h);
    if (!cam || !frame_pose || !kf_pose || n < 0)
        return fail(SVO_ERR_INVALID, "svo_depth_filter_update: bad arguments");
    FilterArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.cam = *cam;
    int* d_n;
    int rc = stage_n(h, n, &d_n);
    if (rc) return rc;
    fa.n_ptr = d_n;
    fa.frame_pose = frame_pose;
    fa.kps2d = const_cast<svo_kp2d*>(kps2d);
    fa.kps3d = kps3d;
    fa.flags = const_cast<uint32_t*>(flags);
    fa.outlier_count = outlier_count; fa.inlier_count = inlier_count;
    fa.kf_inv_depth = kf_inv_depth; fa.kf_variance = kf_variance;
    fa.disparity = disparity; fa.ref3d = ref3d; fa.ref2d = ref2d; fa.kf_pose = kf_pose;
    fa.do_outlier_check = do_outlier_check; fa.do_update = do_update;
    FilterArgs* d;
    rc = stage(h, fa, &d);
    if (rc) return rc;
    launch_filter(d, 1, n, h->stream);
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}
