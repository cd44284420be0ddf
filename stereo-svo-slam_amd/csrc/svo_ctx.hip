// svo_ctx.hip — the tracker behind the C ABI: StereoSlam::new_image
// (src/lib/stereo_slam.cpp:123-271). A group (svo_group, first part of this file) advances B
// sequences together, one kernel launch per stage; the public svo_ctx (end of the file) is a set of
// groups, each on its own stream and host thread, with a queue of submitted frame sets.
//
// Host side = bookkeeping only: image-set pool, argument blocks, the 12-state
// pose Kalman filter (stereo_slam.cpp:296-359) and the keyframe decision. All
// image and keypoint work runs in the kernels of pyramid/sia/klt/reproj/depth/
// keyframe.hip; a tracked frame is nine launches on one stream, one blocking
// read-back of the result block, and (only when a keyframe is due) a second
// batch of five launches.
//
// HBM layout per sequence:
//   image sets  : left halfSample pyramid | right level 0 | Gaussian levels 1,2
//                 (rows padded to 64 B). The current, the previous and every
//                 keyframe's set stay resident (288 GB: ~1 MB per 752x480 set).
//   keypoints   : two SoA sets (KpsDev) ping-ponged by the order-preserving
//                 compactions; per-point scratch (tracked, err, disparity).
//   keyframes   : table of KfDev records + per-keyframe SoA copies.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>
#include <deque>
#include <memory>
#include <string>
#include <cstdlib>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>

#include "../../include/svo_hip.h"
#include "svo_tracker.hpp"

using namespace svo;

int svo_set_error(int code, const char* fmt, ...);   // svo_capi.hip

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return svo_set_error(SVO_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                                 hipGetErrorString(e_), __FILE__, __LINE__);             \
    } while (0)

namespace {

// ------------------------------------------------------ 12-state pose filter
// cv::KalmanFilter(12,12) as configured in the StereoSlam ctor
// (src/lib/stereo_slam.cpp:29-41) and driven by update_pose (:296-359).
// cv::gemm on float data: double accumulation, float store; the gain comes
// out of cv::solve(DECOMP_SVD) (Jacobi SVD, svo_device.hpp).
struct PoseFilter {
    static constexpr int N = 12;
    float statePre[N], statePost[N];
    float A[N * N], Hm[N * N], Q[N * N], R[N * N];
    float errorCovPre[N * N], errorCovPost[N * N], gain[N * N];

    static void identity(float* m, float v) {
        std::memset(m, 0, sizeof(float) * N * N);
        for (int i = 0; i < N; i++) m[i * N + i] = v;
    }
    void init() {
        std::memset(this, 0, sizeof(*this));
        identity(A, 1.f); identity(Hm, 1.f); identity(Q, 100.f); identity(R, 1.f);
        identity(errorCovPost, 1.f);
    }
    static void gemm(const float* a, const float* b, bool bt, double alpha, const float* c,
                     double beta, float* d, int m, int k, int n) {
        float tmp[N * N];
        for (int i = 0; i < m; i++)
            for (int j = 0; j < n; j++) {
                double s = 0;
                for (int p = 0; p < k; p++)
                    s += (double)a[i * k + p] * (double)(bt ? b[j * k + p] : b[p * n + j]);
                s *= alpha;
                if (c) s += (double)c[i * n + j] * beta;
                tmp[i * n + j] = (float)s;
            }
        std::memcpy(d, tmp, sizeof(float) * m * n);
    }
    static void solve_svd(const float* Am, const float* B, float* X) {
        float At[N][N], Vt[N][N], W[N];
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) At[i][j] = Am[j * N + i];
        jacobi_svd<N, N>(At, W, Vt);
        for (int i = 0; i < N * N; i++) X[i] = 0;
        double threshold = 0;
        for (int i = 0; i < N; i++) threshold += W[i];
        threshold *= (float)(DBL_EPSILON * 2);
        for (int i = 0; i < N; i++) {
            double wi = W[i];
            if (std::fabs(wi) <= threshold) continue;
            wi = 1 / wi;
            double buffer[N];
            for (int j = 0; j < N; j++) buffer[j] = 0;
            for (int r = 0; r < N; r++) {
                const float s = At[i][r];
                for (int j = 0; j < N; j++) buffer[j] = buffer[j] + (double)(s * B[r * N + j]);
            }
            for (int j = 0; j < N; j++) buffer[j] *= wi;
            for (int r = 0; r < N; r++) {
                const float s = Vt[i][r];
                for (int j = 0; j < N; j++) X[r * N + j] = (float)(X[r * N + j] + s * buffer[j]);
            }
        }
    }
    void predict() {
        float temp1[N * N];
        gemm(A, statePost, false, 1, nullptr, 0, statePre, N, N, 1);
        gemm(A, errorCovPost, false, 1, nullptr, 0, temp1, N, N, N);
        gemm(temp1, A, true, 1, Q, 1, errorCovPre, N, N, N);
        std::memcpy(statePost, statePre, sizeof(statePre));
        std::memcpy(errorCovPost, errorCovPre, sizeof(errorCovPre));
    }
    void correct(const float* z) {
        float temp2[N * N], temp3[N * N], temp4[N * N], temp5[N], hx[N];
        gemm(Hm, errorCovPre, false, 1, nullptr, 0, temp2, N, N, N);
        gemm(temp2, Hm, true, 1, R, 1, temp3, N, N, N);
        solve_svd(temp3, temp2, temp4);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) gain[i * N + j] = temp4[j * N + i];
        gemm(Hm, statePre, false, 1, nullptr, 0, hx, N, N, 1);
        for (int i = 0; i < N; i++) temp5[i] = z[i] - hx[i];
        gemm(gain, temp5, false, 1, statePre, 1, statePost, N, N, 1);
        gemm(gain, temp2, false, -1, errorCovPre, 1, errorCovPost, N, N, N);
    }
    // StereoSlam::update_pose
    void update(const float pose[6], const float speed[6], const float pv[6], const float sv[6],
                double dt, float filtered[6]) {
        for (int i = 0; i < 6; i++) A[i * N + 6 + i] = (float)dt;
        predict();
        for (int i = 0; i < 6; i++) { R[i * N + i] = pv[i]; R[(6 + i) * N + 6 + i] = sv[i]; }
        float z[N];
        for (int i = 0; i < 6; i++) { z[i] = pose[i]; z[6 + i] = speed[i]; }
        correct(z);
        for (int i = 0; i < 6; i++) filtered[i] = statePost[i];
    }
};

struct ImageSet {
    uint8_t* base = nullptr;
    ImgView left[SVO_MAX_PYRAMID_LEVELS];
    ImgView right;
    ImgView lk[SVO_LK_LEVELS];
    ImgView own_left0, own_right;     // the set's own level-0 storage (left[0] / right alias the caller's
                                      // images instead with SVO_MEM_DEVICE_BORROW)
    int refs = 0;
};

struct FrameResult {            // device -> host, one per sequence and frame
    float pose_sia[6];
    float pose_refined[6];
    float sia_cost, reproj_cost;
    int inside, overflow, kf_n, old_count;
    int min_kf;                 // smallest origin-keyframe id of the frame's keypoints (compact_kernel) ...
    unsigned live_kf[2];        // ... and which of the 64 keyframes from there on still have keypoints in the frame
    svo_gn_trace sia_trace[SVO_MAX_PYRAMID_LEVELS];
    svo_gn_trace reproj_trace;
};

struct KfHost {
    ImageSet* set;
    float pose[6];
    int n;
    svo_kp2d* kps2d; svo_kp3d* kps3d; uint32_t* flags; int* outl; int* inl;   // device
    int* kf_id; int* kp_index; float* score; int* level_type; uint32_t* color; float* kfx; float* kfP;
};

struct Seq {
    KpsDev kps[2];
    int cur = 0;
    int* d_n = nullptr;          // [2] keypoint counts of the two sets
    svo_kp2d* tracked = nullptr;
    float* klt_err = nullptr;
    uint8_t* klt_status = nullptr;
    float* disparity = nullptr;
    float* sia_rec = nullptr;        // per-level alignment records (sia_prep_kernel)
    float* sia_kpws = nullptr;
    PoseMats* sia_mats = nullptr;    // rotation matrices of the aligned pose (sia_gn_kernel -> klt_track_kernel)
    uint8_t* tmpl_base = nullptr;    // KLT template cache: tmpl_kf blocks (a ring over the sequence's keyframes)
    uint8_t* tmpl_valid = nullptr;   // their "stored" flags
    KfDev* d_kfs = nullptr;
    std::vector<KfHost> kfs;
    int kfs_retired = 0;             // keyframes [0, kfs_retired) have given their image sets back
    DetCell* det = nullptr; int* n_det = nullptr;
    DetCell* sel = nullptr; int* sel_level = nullptr; int* sel_cell = nullptr; int* occupied = nullptr;
    uint32_t* color_lcg = nullptr;
    std::vector<ImageSet*> free_sets;
    ImageSet* cur_set = nullptr;
    ImageSet* prev_set = nullptr;
    // host state
    PoseFilter kf;
    int frame_id = -1;
    double ts = 0;
    float pose[6] = {0, 0, 0, 0, 0, 0};
    std::vector<svo_pose> trajectory;
    svo_frame_stats stats;
    int n_host = 0;
    // pose-filter update of the last frame, deferred so that it overlaps the next frame's kernels
    bool pending = false;
    float pending_pose[6] = {0, 0, 0, 0, 0, 0};
    double pending_ts = 0;
};

// Host worker pool for the per-sequence host work of a step (pose filter, argument blocks):
// sequences are independent, and at 256 sequences the 12-state filter alone (a 12x12 Jacobi SVD
// per sequence) costs as much host time as the GPU needs for the whole frame. The calling
// thread takes part; workers sleep between steps.
class HostPool {
public:
    explicit HostPool(int n_workers) {
        for (int i = 0; i < n_workers; i++) workers_.emplace_back([this] { run(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // fn(i) for i in [0, n), in chunks; returns when all are done
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (workers_.empty() || n < 16) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; n_ = n; next_.store(0); busy_ = (int)workers_.size(); gen_++;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return busy_ == 0; });
        fn_ = nullptr;
    }
private:
    void drain() {
        for (;;) {
            const int i0 = next_.fetch_add(kChunk);
            if (i0 >= n_) break;
            const int i1 = std::min(n_, i0 + kChunk);
            for (int i = i0; i < i1; i++) (*fn_)(i);
        }
    }
    void run() {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            drain();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--busy_ == 0) done_.notify_one();
            }
        }
    }
    static constexpr int kChunk = 4;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, busy_ = 0;
    unsigned gen_ = 0;
    bool stop_ = false;
};

}  // namespace

struct svo_group {
    int device, B, width, height, cap, rec_cap, max_kf, n_lk, det_levels, max_cells, merge_cells;
    svo_camera_settings cam;
    hipStream_t stream;
    std::vector<Seq> seqs;
    // argument blocks: pinned host mirror + device copy, one array per kernel
    uint8_t* h_args = nullptr; uint8_t* d_args = nullptr; size_t args_bytes = 0, frame_args_bytes = 0;
    size_t off_hs, off_lk, off_compact, off_sia, off_klt, off_rp, off_ssd, off_filt, off_det,
        off_merge, off_init, off_guess, off_enable, off_kfdev;
    FrameResult* d_res = nullptr; FrameResult* h_res = nullptr;
    int* h_n = nullptr;          // pinned [B*2]
    int* d_n_all = nullptr;      // [B*2]
    // d_res | d_n_all | d_inside are one device block mirrored by one pinned block: the end-of-frame
    // read-back is a single copy, the keyframe decision reads back only the B inside-counters
    int* d_inside = nullptr; int* h_inside = nullptr;
    // host-resident input frames land here first (2 x B frames; runs of contiguous frames as one
    // copy) and are then ingested like device-resident ones
    uint8_t* d_stage_in = nullptr; size_t stage_frame_bytes = 0;
    size_t readback_bytes = 0;
    bool timing = false;
    bool failed = false;
    int exact_pinv = 1;          // reference-order Gauss-Newton unless svo_ctx_set_fast_solver(ctx, 1)
    hipEvent_t ev[10] = {nullptr};
    size_t set_bytes = 0;
    std::vector<void*> allocs;   // everything to free
    std::vector<uint8_t*> kf_slabs;   // free per-keyframe keypoint storage (allocated in chunks)
    std::vector<uint8_t*> set_slabs;  // free image-set storage (allocated in chunks)
    // KLT template cache (klt.hip): the templates of a keyframe's keypoints stay in HBM while the keyframe is one
    // of the last tmpl_kf of its sequence (0: off)
    int tmpl_kf = 0, tmpl_cap = 0;
    size_t tmpl_block_bytes = 0, tmpl_valid_bytes = 0;
    svo_totals totals;
    bool retire_kf_images = true;    // SVO_KEEP_KEYFRAME_IMAGES=1: keep every keyframe's image set (the reference's behaviour)
    int image_sets = 0;              // image sets allocated so far
    HostPool* pool = nullptr;
    double host_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // SVO_HOST_TIMING diagnostic: host phases of a step
    long host_steps = 0;
};

namespace {

template <typename T>
int dev_alloc(svo_group* c, T** p, size_t count) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, sizeof(T) * std::max<size_t>(count, 1)));
    HIP_TRY(hipMemset(q, 0, sizeof(T) * std::max<size_t>(count, 1)));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return SVO_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Image-set storage comes from slabs allocated in chunks: one hipMalloc (a device-wide
// synchronising call) per chunk of sets, not per set — every keyframe keeps its set for good, so a
// long run asks for one per keyframe.
int grow_set_slabs(svo_group* c, int count) {
    uint8_t* base = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&base), c->set_bytes * (size_t)count));
    c->allocs.push_back(base);
    for (int i = count - 1; i >= 0; i--) c->set_slabs.push_back(base + c->set_bytes * (size_t)i);
    return SVO_OK;
}

// layout of one image set (byte offsets into its slab); fills c->set_bytes
void image_set_layout(svo_group* c, ImageSet* s, size_t* offs_left, size_t* offs_lk, size_t* off_right) {
    size_t off = 0;
    int w = c->width, h = c->height;
    for (int l = 0; l < c->cam.max_pyramid_levels; l++) {
        const int stride = (int)align_up((size_t)std::max(w, 1), 64);
        offs_left[l] = off;
        s->left[l] = ImgView{nullptr, w, h, stride};
        off += align_up((size_t)stride * std::max(h, 1), 256);
        w /= 2; h /= 2;
    }
    {
        const int stride = (int)align_up((size_t)c->width, 64);
        *off_right = off;
        s->right = ImgView{nullptr, c->width, c->height, stride};
        off += align_up((size_t)stride * c->height, 256);
    }
    w = c->width; h = c->height;
    for (int l = 1; l < c->n_lk; l++) {
        w = (w + 1) / 2; h = (h + 1) / 2;
        const int stride = (int)align_up((size_t)w, 64);
        offs_lk[l] = off;
        s->lk[l] = ImgView{nullptr, w, h, stride};
        off += align_up((size_t)stride * h, 256);
    }
    c->set_bytes = off;
}

int new_image_set(svo_group* c, ImageSet** out) {
    ImageSet* s = new ImageSet();
    c->image_sets++;
    size_t offs_left[SVO_MAX_PYRAMID_LEVELS], offs_lk[SVO_LK_LEVELS], off_right;
    image_set_layout(c, s, offs_left, offs_lk, &off_right);
    if (c->set_slabs.empty()) {
        const int rc = grow_set_slabs(c, std::max(c->B, 16));
        if (rc) { delete s; return rc; }
    }
    s->base = c->set_slabs.back();
    c->set_slabs.pop_back();
    for (int l = 0; l < c->cam.max_pyramid_levels; l++) s->left[l].data = s->base + offs_left[l];
    s->right.data = s->base + off_right;
    s->lk[0] = s->left[0];
    s->own_left0 = s->left[0];
    s->own_right = s->right;
    for (int l = 1; l < c->n_lk; l++) s->lk[l].data = s->base + offs_lk[l];
    *out = s;
    return SVO_OK;
}

int acquire_set(svo_group* c, Seq& q, ImageSet** out) {
    if (q.free_sets.empty()) {
        ImageSet* s;
        int rc = new_image_set(c, &s);
        if (rc) return rc;
        q.free_sets.push_back(s);
    }
    *out = q.free_sets.back();
    q.free_sets.pop_back();
    (*out)->refs = 1;
    return SVO_OK;
}

void release_set(Seq& q, ImageSet* s) {
    if (!s) return;
    if (--s->refs <= 0) q.free_sets.push_back(s);
}

int alloc_kps(svo_group* c, KpsDev& k, int* n_ptr) {
    const size_t cap = c->cap;
    int rc;
    if ((rc = dev_alloc(c, &k.kps2d, cap))) return rc;
    if ((rc = dev_alloc(c, &k.kps3d, cap))) return rc;
    if ((rc = dev_alloc(c, &k.flags, cap))) return rc;
    if ((rc = dev_alloc(c, &k.kf_id, cap))) return rc;
    if ((rc = dev_alloc(c, &k.kp_index, cap))) return rc;
    if ((rc = dev_alloc(c, &k.outl, cap))) return rc;
    if ((rc = dev_alloc(c, &k.inl, cap))) return rc;
    if ((rc = dev_alloc(c, &k.kfx, cap))) return rc;
    if ((rc = dev_alloc(c, &k.kfP, cap))) return rc;
    if ((rc = dev_alloc(c, &k.score, cap))) return rc;
    if ((rc = dev_alloc(c, &k.level_type, cap))) return rc;
    if ((rc = dev_alloc(c, &k.color, cap))) return rc;
    k.n = n_ptr;
    return SVO_OK;
}

template <typename T>
T* args_at(svo_group* c, size_t off, int s) { return reinterpret_cast<T*>(c->h_args + off) + s; }
template <typename T>
T* dargs_at(svo_group* c, size_t off, int s = 0) { return reinterpret_cast<T*>(c->d_args + off) + s; }

// per-keyframe keypoint storage: 15 dwords per keypoint. Slabs come from chunks of `count`
// (one hipMalloc — a device-wide synchronising call — per chunk, not per keyframe).
size_t kf_slab_bytes(const svo_group* c) { return align_up((size_t)c->cap * 15 * 4, 256); }

int grow_kf_slabs(svo_group* c, int count) {
    const size_t sb = kf_slab_bytes(c);
    uint8_t* base = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&base), sb * count));
    c->allocs.push_back(base);
    for (int i = count - 1; i >= 0; i--) c->kf_slabs.push_back(base + sb * i);
    return SVO_OK;
}

int new_keyframe_storage(svo_group* c, Seq& q, int s, int id) {
    if (id >= c->max_kf) return svo_set_error(SVO_ERR_CAPACITY, "more than %d keyframes", c->max_kf);
    KfHost k;
    std::memset(&k, 0, sizeof(k));
    const size_t cap = c->cap;
    if (c->kf_slabs.empty()) {
        const int rc = grow_kf_slabs(c, std::max(c->B, 32));
        if (rc) return rc;
    }
    uint8_t* base = c->kf_slabs.back();
    c->kf_slabs.pop_back();
    k.kps3d = reinterpret_cast<svo_kp3d*>(base);
    k.kps2d = reinterpret_cast<svo_kp2d*>(base + cap * sizeof(svo_kp3d));
    k.flags = reinterpret_cast<uint32_t*>(base + cap * (sizeof(svo_kp3d) + sizeof(svo_kp2d)));
    k.outl = reinterpret_cast<int*>(k.flags + cap);
    k.inl = k.outl + cap;
    k.kf_id = k.inl + cap;
    k.kp_index = k.kf_id + cap;
    k.score = reinterpret_cast<float*>(k.kp_index + cap);
    k.level_type = reinterpret_cast<int*>(k.score + cap);
    k.color = reinterpret_cast<uint32_t*>(k.level_type + cap);
    k.kfx = reinterpret_cast<float*>(k.color + cap);
    k.kfP = k.kfx + cap;
    k.set = q.cur_set;
    q.cur_set->refs++;
    q.kfs.push_back(k);
    KfDev& d = *args_at<KfDev>(c, c->off_kfdev, s);   // pinned staging, stable until the frame ends
    std::memset(&d, 0, sizeof(d));
    for (int l = 0; l < c->n_lk; l++) d.lk[l] = q.cur_set->lk[l];
    d.n_lk = c->n_lk;
    d.kps2d = k.kps2d; d.kps3d = k.kps3d; d.flags = k.flags; d.outlier_count = k.outl; d.inlier_count = k.inl;
    d.kf_id = k.kf_id; d.kp_index = k.kp_index; d.score = k.score; d.level_type = k.level_type;
    d.color = k.color; d.kfx = k.kfx; d.kfP = k.kfP;
    if (c->tmpl_kf > 0) {
        // the keyframe takes the oldest block of the sequence's ring; kf_init_kernel clears the flags and
        // takes the cache away from the keyframe that held the block (id - tmpl_kf: its points are tracked
        // from the images again)
        const int r = id % c->tmpl_kf;
        d.tmpl = q.tmpl_base + (size_t)r * c->tmpl_block_bytes;
        d.tmpl_valid = q.tmpl_valid + (size_t)r * c->tmpl_valid_bytes;
        d.tmpl_cap = c->tmpl_cap;
        d.tmpl_win = c->cam.window_size_opt_flow;
    }
    // (the record reaches the device inside the KfInitArgs block: no copy per keyframe)
    return SVO_OK;
}

}  // namespace

// motion + 12-state filter + trajectory of the last frame (stereo_slam.cpp:250-270).
// Deferred: the next frame's pose guess only needs the state BEFORE this update
// (kf.statePre after its predict() equals the current statePost, dt = 0), so the
// host runs it while the GPU already works on the next frame.
static void flush_one(Seq& q) {
    if (!q.pending) return;
    q.pending = false;
    float prev_pose[6];
    std::memcpy(prev_pose, q.pose, sizeof(prev_pose));
    std::memcpy(q.pose, q.pending_pose, sizeof(q.pose));
    const double dt = q.pending_ts - q.ts;
    const double inv = 1. / dt;
    float motion[6];
    for (int i = 0; i < 6; i++) motion[i] = (float)((q.pose[i] - prev_pose[i]) * inv);
    const float pv[6] = {0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f};
    const float mv[6] = {1, 1, 1, 1, 1, 1};
    float filtered[6];
    q.kf.update(q.pose, motion, pv, mv, 0.0, filtered);
    std::memcpy(q.pose, filtered, sizeof(q.pose));
    q.ts = q.pending_ts;
    svo_pose p;
    std::memcpy(&p, q.pose, sizeof(p));
    q.trajectory.push_back(p);
}
static void flush_pending(svo_group* c) {
    if (c->pool) c->pool->parallel_for((int)c->seqs.size(), [c](int s) { flush_one(c->seqs[s]); });
    else for (Seq& q : c->seqs) flush_one(q);
}

static int grp_create(const svo_camera_settings* cam, int width, int height, int n_sequences,
                              int device, svo_group** out) {
    if (!cam || !out || width < 16 || height < 16 || n_sequences < 1)
        return svo_set_error(SVO_ERR_INVALID, "svo_ctx_create: bad arguments");
    if (cam->max_pyramid_levels < 1 || cam->max_pyramid_levels > 7 ||
        cam->min_pyramid_level_pose_estimation < 0 ||
        cam->min_pyramid_level_pose_estimation >= cam->max_pyramid_levels)
        return svo_set_error(SVO_ERR_INVALID, "max_pyramid_levels must be 1..7 and > min level");
    if (cam->window_size_opt_flow < 3 || cam->window_size_opt_flow > 35 ||
        cam->window_size_depth_calculator < 1 || cam->window_size_depth_calculator > 35 ||
        cam->search_x < 0 || cam->search_x > 64 || cam->search_y < 0 || cam->search_y > 8)
        return svo_set_error(SVO_ERR_INVALID, "windows <= 35, search_x <= 64, search_y <= 8 supported");
    if (cam->window_size_pose_estimator != 4)   // PATCH_SIZE, src/lib/pose_estimator.cpp:68
        return svo_set_error(SVO_ERR_INVALID, "window_size_pose_estimator must be 4");
    if (cam->grid_width < 4 || cam->grid_height < 4 || cam->grid_width > 96 || cam->grid_height > 64)
        return svo_set_error(SVO_ERR_INVALID, "grid cell must be within 4..96 x 4..64");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return svo_set_error(SVO_ERR_NO_DEVICE, "no HIP device visible: libsvo_hip has no CPU fallback");
    if (device < 0 || device >= count) return svo_set_error(SVO_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    svo_group* c = new (std::nothrow) svo_group();
    if (!c) return svo_set_error(SVO_ERR_INVALID, "out of host memory");
    c->device = device; c->B = n_sequences; c->width = width; c->height = height; c->cam = *cam;
    std::memset(&c->totals, 0, sizeof(c->totals));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const int cells = (width / cam->grid_width) * (height / cam->grid_height);
    c->cap = (int)align_up((size_t)(2 * cells + 128), 64);
    c->rec_cap = (int)align_up((size_t)c->cap, 512);   // whole passes of the widest alignment workgroup
    c->max_kf = 4096;
    if (const char* e = std::getenv("SVO_KEEP_KEYFRAME_IMAGES")) c->retire_kf_images = std::atoi(e) == 0;
    // usable LK levels (cv::buildOpticalFlowPyramid stops at levels not larger than the window)
    {
        int n = SVO_LK_LEVELS, w = width, h = height;
        for (int l = 0; l < SVO_LK_LEVELS; l++) {
            w = (w + 1) / 2; h = (h + 1) / 2;
            if (w <= cam->window_size_opt_flow || h <= cam->window_size_opt_flow) { n = l + 1; break; }
        }
        c->n_lk = n;
    }
    c->det_levels = cam->max_pyramid_levels / 2;
    c->max_cells = 1;
    for (int l = 0; l < c->det_levels; l++) {
        const int gw = cam->grid_width >> l, gh = cam->grid_height >> l;
        if (gw <= 0 || gh <= 0) { c->det_levels = l; break; }
        const int nc = ((width >> l) / gw) * std::max((height >> l) / gh, 1);
        c->max_cells = std::max(c->max_cells, nc);
    }
    c->merge_cells = ((width + cam->grid_height - 1) / cam->grid_height) *
                     ((height + cam->grid_width - 1) / cam->grid_width);

    const int B = c->B;
    // argument blocks
    size_t off = 0;
    auto reserve = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    c->off_hs = reserve(sizeof(PyrArgs) * B);
    c->off_compact = reserve(sizeof(CompactArgs) * B);
    c->off_sia = reserve(sizeof(SiaArgs) * B);
    c->off_klt = reserve(sizeof(KltArgs) * B);
    c->off_rp = reserve(sizeof(ReprojArgs) * B);
    c->off_ssd = reserve(sizeof(SsdArgs) * B);
    c->off_filt = reserve(sizeof(FilterArgs) * B);
    c->off_guess = reserve(sizeof(float) * 8 * B);
    c->frame_args_bytes = off;                    // everything a tracked frame uploads; the rest is keyframe-only
    c->off_det = reserve(sizeof(DetectArgs) * B);
    c->off_merge = reserve(sizeof(MergeArgs) * B);
    c->off_init = reserve(sizeof(KfInitArgs) * B);
    c->off_enable = reserve(sizeof(int) * B);
    c->off_kfdev = reserve(sizeof(KfDev) * B);
    c->args_bytes = off;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_args), off, hipHostMallocDefault));
    std::memset(c->h_args, 0, off);
    int rc;
    if ((rc = dev_alloc(c, &c->d_args, off))) return rc;
    {
        const size_t res_bytes = sizeof(FrameResult) * B, n_bytes = sizeof(int) * 2 * B, in_bytes = sizeof(int) * B;
        c->readback_bytes = res_bytes + n_bytes;
        uint8_t* hb = nullptr; uint8_t* db = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&hb), res_bytes + n_bytes + in_bytes, hipHostMallocDefault));
        std::memset(hb, 0, res_bytes + n_bytes + in_bytes);
        if ((rc = dev_alloc(c, &db, res_bytes + n_bytes + in_bytes))) return rc;
        c->h_res = reinterpret_cast<FrameResult*>(hb); c->d_res = reinterpret_cast<FrameResult*>(db);
        c->h_n = reinterpret_cast<int*>(hb + res_bytes); c->d_n_all = reinterpret_cast<int*>(db + res_bytes);
        c->h_inside = reinterpret_cast<int*>(hb + res_bytes + n_bytes);
        c->d_inside = reinterpret_cast<int*>(db + res_bytes + n_bytes);
    }
    for (int i = 0; i < 10; i++) HIP_TRY(hipEventCreate(&c->ev[i]));

    c->seqs.resize(B);
    {   // the first four image sets of every sequence: one allocation
        ImageSet probe;
        size_t ol[SVO_MAX_PYRAMID_LEVELS], olk[SVO_LK_LEVELS], orr;
        image_set_layout(c, &probe, ol, olk, &orr);
        if ((rc = grow_set_slabs(c, 4 * B))) return rc;
    }
    if (B >= 16) {
        // SVO_HOST_THREADS: host threads per group for the deferred pose filter
        // (default 1 = off: at the measured kernel times the filter hides behind the GPU work)
        int nt = 1;
        if (const char* e = std::getenv("SVO_HOST_THREADS")) nt = std::max(1, std::atoi(e));
        if (nt > 1) c->pool = new HostPool(nt - 1);
    }
    for (int s = 0; s < B; s++) {
        Seq& q = c->seqs[s];
        q.d_n = c->d_n_all + 2 * s;
        if ((rc = alloc_kps(c, q.kps[0], q.d_n))) return rc;
        if ((rc = alloc_kps(c, q.kps[1], q.d_n + 1))) return rc;
        if ((rc = dev_alloc(c, &q.tracked, (size_t)c->cap))) return rc;
        if ((rc = dev_alloc(c, &q.klt_err, (size_t)c->cap))) return rc;
        if ((rc = dev_alloc(c, &q.klt_status, (size_t)c->cap))) return rc;
        if ((rc = dev_alloc(c, &q.disparity, (size_t)c->cap))) return rc;
        if ((rc = dev_alloc(c, &q.sia_rec, sia_rec_ws_floats(*cam, c->rec_cap)))) return rc;
        if ((rc = dev_alloc(c, &q.sia_kpws, (size_t)9 * c->rec_cap))) return rc;
        if ((rc = dev_alloc(c, &q.sia_mats, 1))) return rc;
        if ((rc = dev_alloc(c, &q.d_kfs, (size_t)c->max_kf))) return rc;
        if ((rc = dev_alloc(c, &q.det, (size_t)SVO_MAX_PYRAMID_LEVELS * c->max_cells))) return rc;
        if ((rc = dev_alloc(c, &q.n_det, (size_t)SVO_MAX_PYRAMID_LEVELS))) return rc;
        if ((rc = dev_alloc(c, &q.sel, (size_t)c->max_cells))) return rc;
        if ((rc = dev_alloc(c, &q.sel_level, (size_t)c->max_cells))) return rc;
        if ((rc = dev_alloc(c, &q.sel_cell, (size_t)c->max_cells))) return rc;
        if ((rc = dev_alloc(c, &q.occupied, (size_t)c->merge_cells))) return rc;
        if ((rc = dev_alloc(c, &q.color_lcg, (size_t)1))) return rc;
        const uint32_t lcg = 12345u;
        HIP_TRY(hipMemcpy(q.color_lcg, &lcg, sizeof(lcg), hipMemcpyHostToDevice));
        q.kf.init();
        std::memset(&q.stats, 0, sizeof(q.stats));
        for (int i = 0; i < 4; i++) {       // pre-allocate a few image sets
            ImageSet* is;
            if ((rc = new_image_set(c, &is))) return rc;
            q.free_sets.push_back(is);
        }
    }
    if ((rc = grow_kf_slabs(c, std::max(2 * B, 32)))) return rc;   // the first keyframes never allocate
    {
        // KLT template cache: SVO_KLT_CACHE_KF keyframes per sequence (default 8 while a keyframe's block stays below
        // 8 MB, else 4; 0 = off), as many as fit a third of the free device memory. On closed camera loops every
        // keyframe keeps keypoints in view, and those of keyframes that have left the ring build their templates
        // on every frame: 8 instead of 4 blocks per sequence are +0.8 % frames/s at C2 (profiles/r03_ab_steps.txt).
        // A keypoint index beyond tmpl_cap (more points than grid cells + 64 in the frame that made the keyframe)
        // is tracked without the cache.
        const int cells_ = (width / cam->grid_width) * (height / cam->grid_height);
        c->tmpl_cap = std::min(c->cap, cells_ + 64);
        c->tmpl_block_bytes = align_up((size_t)c->tmpl_cap * SVO_LK_LEVELS * klt_template_bytes(cam->window_size_opt_flow), 256);
        int K = c->tmpl_block_bytes <= ((size_t)8 << 20) ? 8 : 4;
        if (const char* e = std::getenv("SVO_KLT_CACHE_KF")) K = std::max(0, std::min(std::atoi(e), 64));
        c->tmpl_valid_bytes = align_up((size_t)c->tmpl_cap * SVO_LK_LEVELS, 256);
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        while (K > 0 && (size_t)B * K * c->tmpl_block_bytes > free_b / 3) K--;
        c->tmpl_kf = K;
        if (K > 0) {
            uint8_t* base = nullptr; uint8_t* vbase = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&base), (size_t)B * K * c->tmpl_block_bytes));
            c->allocs.push_back(base);
            if ((rc = dev_alloc(c, &vbase, (size_t)B * K * c->tmpl_valid_bytes))) return rc;
            for (int s = 0; s < B; s++) {
                c->seqs[s].tmpl_base = base + (size_t)s * K * c->tmpl_block_bytes;
                c->seqs[s].tmpl_valid = vbase + (size_t)s * K * c->tmpl_valid_bytes;
            }
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    *out = c;
    return SVO_OK;
}

static int grp_destroy(svo_group* c) {
    if (!c) return SVO_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (std::getenv("SVO_HOST_TIMING") && c->host_steps > 0) {
        static const char* names[7] = {"args", "launch", "pose_filter", "wait_frame", "kf_enqueue", "wait_kf", "bookkeeping"};
        std::fprintf(stderr, "[svo host ms/step over %ld steps]", c->host_steps);
        for (int i = 0; i < 7; i++) std::fprintf(stderr, " %s=%.3f", names[i], c->host_ms[i] / c->host_steps);
        std::fprintf(stderr, "\n");
    }
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->h_args) (void)hipHostFree(c->h_args);
    if (c->h_res) (void)hipHostFree(c->h_res);   // one pinned block: results, counts, inside counters
    for (int i = 0; i < 10; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    // ImageSet structs: owned by the free lists, the current/previous pointers and keyframes
    for (Seq& q : c->seqs) {
        std::vector<ImageSet*> all(q.free_sets);
        if (q.cur_set) all.push_back(q.cur_set);
        if (q.prev_set) all.push_back(q.prev_set);
        for (auto& k : q.kfs)
            if (k.set) all.push_back(k.set);
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        for (ImageSet* s : all) delete s;
    }
    (void)hipStreamDestroy(c->stream);
    delete c->pool;
    delete c;
    return SVO_OK;
}

static int grp_set_exact_pinv(svo_group* c, int on) {
    if (!c) return svo_set_error(SVO_ERR_INVALID, "null ctx");
    c->exact_pinv = on != 0;
    return SVO_OK;
}

static int grp_enable_timing(svo_group* c, int on) {
    if (!c) return svo_set_error(SVO_ERR_INVALID, "null ctx");
    c->timing = on != 0;
    return SVO_OK;
}

// keyframe creation for the sequences flagged in `need`: their argument blocks are
// packed into the first m slots, so the five launches cover exactly those sequences
static int enqueue_keyframes(svo_group* c, const std::vector<int>& need, bool first_frame) {
    const int B = c->B;
    int m = 0;
    for (int s = 0; s < B; s++) {
        if (!need[s]) continue;
        Seq& q = c->seqs[s];
        const int slot = m++;
        const int id = (int)q.kfs.size();
        int rc = new_keyframe_storage(c, q, s, id);
        if (rc) return rc;
        // find_bad_keypoints: cur -> other, then the other set is current
        CompactArgs* ca = args_at<CompactArgs>(c, c->off_compact, slot);
        std::memset(ca, 0, sizeof(*ca));
        ca->src = q.kps[q.cur]; ca->dst = q.kps[q.cur ^ 1]; ca->mode = 1;
        ca->width = c->width; ca->height = c->height;
        q.cur ^= 1;
        DetectArgs* da = args_at<DetectArgs>(c, c->off_det, slot);
        std::memset(da, 0, sizeof(*da));
        for (int l = 0; l < c->cam.max_pyramid_levels; l++) da->level[l] = q.cur_set->left[l];
        da->n_levels = c->det_levels; da->grid_w = c->cam.grid_width; da->grid_h = c->cam.grid_height;
        da->out = q.det; da->n_out = q.n_det; da->max_cells = c->max_cells;
        MergeArgs* ma = args_at<MergeArgs>(c, c->off_merge, slot);
        std::memset(ma, 0, sizeof(*ma));
        ma->cam = c->cam; ma->width = c->width; ma->height = c->height;
        ma->det = q.det; ma->n_det = q.n_det; ma->n_levels = c->det_levels; ma->max_cells = c->max_cells;
        ma->kps = q.kps[q.cur]; ma->cap = c->cap;
        ma->sel = q.sel; ma->sel_level = q.sel_level; ma->sel_cell = q.sel_cell; ma->occupied = q.occupied;
        ma->old_count = &c->d_res[s].old_count; ma->overflow = &c->d_res[s].overflow;
        SsdArgs* sa = args_at<SsdArgs>(c, c->off_ssd, slot);
        std::memset(sa, 0, sizeof(*sa));
        sa->left = q.cur_set->left[0]; sa->right = q.cur_set->right;
        sa->n_ptr = q.kps[q.cur].n; sa->kps2d = q.kps[q.cur].kps2d; sa->disparity = q.disparity;
        sa->win = c->cam.window_size_depth_calculator; sa->search_x = c->cam.search_x;
        sa->search_y = c->cam.search_y; sa->clamp_half = 0;
        sa->first = 0; sa->first_ptr = &c->d_res[s].old_count;
        KfInitArgs* ia = args_at<KfInitArgs>(c, c->off_init, slot);
        std::memset(ia, 0, sizeof(*ia));
        ia->cam = c->cam; ia->kps = q.kps[q.cur]; ia->old_count = &c->d_res[s].old_count;
        ia->disparity = q.disparity; ia->frame_pose = c->d_res[s].pose_refined;
        ia->first_frame = first_frame ? 1 : 0; ia->new_kf_id = id; ia->kfs = q.d_kfs;
        ia->color_lcg = q.color_lcg; ia->n_out = &c->d_res[s].kf_n;
        ia->record = *args_at<KfDev>(c, c->off_kfdev, s);
        ia->tmpl_valid_bytes = (int)c->tmpl_valid_bytes;
        ia->evict_id = (c->tmpl_kf > 0 && id >= c->tmpl_kf) ? id - c->tmpl_kf : -1;
        ca->zero = q.n_det; ca->zero_count = SVO_MAX_PYRAMID_LEVELS;      // (detection counters: cleared by the compaction kernel)
    }
    if (m == 0) return SVO_OK;
    HIP_TRY(hipMemcpyAsync(c->d_args, c->h_args, c->args_bytes, hipMemcpyHostToDevice, c->stream));
    // (a launch that fails must not be masked by the next one that succeeds: checked one by one)
    launch_compact(dargs_at<CompactArgs>(c, c->off_compact), m, c->cap, c->stream);
    HIP_TRY(hipGetLastError());
    if (c->det_levels > 0) {
        launch_detect(dargs_at<DetectArgs>(c, c->off_det), m, c->max_cells, c->det_levels, c->cam.grid_width, c->cam.grid_height, c->stream);
        HIP_TRY(hipGetLastError());
    }
    launch_select_merge(dargs_at<MergeArgs>(c, c->off_merge), m, c->max_cells, c->stream);
    HIP_TRY(hipGetLastError());
    launch_ssd(dargs_at<SsdArgs>(c, c->off_ssd), m, c->cap, c->cam.window_size_depth_calculator, c->cam.search_y, c->stream);
    HIP_TRY(hipGetLastError());
    launch_kf_init(dargs_at<KfInitArgs>(c, c->off_init), m, c->stream);
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}

static int grp_new_images_impl(svo_group* c, const uint8_t* const* left, const uint8_t* const* right,
                                  int stride, const float* time_stamps, int mem) {
    HIP_TRY(hipSetDevice(c->device));
    const auto wall0 = std::chrono::steady_clock::now();
    const int B = c->B;
    const bool first = c->seqs[0].frame_id < 0;
    int rc;
    // Sequences whose image pointers are NULL sit this step out (their state is untouched): a ctx
    // can hold sequences of different lengths. The others are packed into the first M slots of
    // every argument array, so the launches cover exactly them.
    std::vector<int> act;
    act.reserve(B);
    for (int s = 0; s < B; s++)
        if (left[s] && right[s]) act.push_back(s);
        else if ((left[s] != nullptr) != (right[s] != nullptr))
            return svo_set_error(SVO_ERR_INVALID, "svo_new_images: sequence %d has only one image", s);
    const int M = (int)act.size();
    if (first && M != B) return svo_set_error(SVO_ERR_INVALID, "svo_new_images: the first frame needs every sequence");
    if (M == 0) return SVO_OK;
#define SVO_MARK(i) do { if (c->timing) HIP_TRY(hipEventRecord(c->ev[i], c->stream)); } while (0)
    auto hclock = wall0;
    auto hlap = [&](int i) {
        const auto now = std::chrono::steady_clock::now();
        c->host_ms[i] += std::chrono::duration<double, std::milli>(now - hclock).count();
        hclock = now;
    };
    SVO_MARK(0);

    // ---- images in, pyramids
    if (mem == SVO_MEM_HOST) {
        const size_t used = (size_t)(c->height - 1) * stride + c->width;      // bytes of one frame that are read
        const size_t fb = align_up((size_t)c->height * stride, 256);
        if (fb > c->stage_frame_bytes) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if ((rc = dev_alloc(c, &c->d_stage_in, fb * 2 * B))) return rc;    // (an outgrown buffer is freed with the ctx)
            c->stage_frame_bytes = fb;
        }
        // slots 0..B-1: left frames, B..2B-1: right frames. Host frames that follow each other at
        // exactly one frame's distance (one [B][H][stride] block per side) go as ONE 2D copy:
        // a "row" is a whole frame
        const size_t spacing = (size_t)c->height * stride;
        for (int side = 0; side < 2; side++) {
            const uint8_t* const* src = side ? right : left;
            int s0 = 0;
            while (s0 < B) {
                if (!src[s0]) { s0++; continue; }
                int s1 = s0 + 1;
                while (s1 < B && src[s1] && src[s1] == src[s1 - 1] + spacing) s1++;
                uint8_t* dst = c->d_stage_in + (size_t)(side * B + s0) * c->stage_frame_bytes;
                if (s1 - s0 > 1) {
                    HIP_TRY(hipMemcpy2DAsync(dst, c->stage_frame_bytes, src[s0], spacing, spacing, s1 - s0,
                                             hipMemcpyHostToDevice, c->stream));
                } else {
                    HIP_TRY(hipMemcpyAsync(dst, src[s0], used, hipMemcpyHostToDevice, c->stream));
                }
                s0 = s1;
            }
        }
    }
    int pyr_stream = -1;               // row block of the row-streaming pyramid kernel, 0: some frame of the step does not fit it
    for (int j = 0; j < M; j++) {
        const int s = act[j];
        Seq& q = c->seqs[s];
        release_set(q, q.prev_set);
        q.prev_set = q.cur_set;
        if ((rc = acquire_set(c, q, &q.cur_set))) return rc;
        ImageSet* is = q.cur_set;
        PyrArgs* hs = args_at<PyrArgs>(c, c->off_hs, j);
        std::memset(hs, 0, sizeof(*hs));
        hs->n_levels = c->cam.max_pyramid_levels;
        if (mem == SVO_MEM_DEVICE_BORROW) {
            // level 0 of both pyramids and the right image ARE the caller's images (like the
            // reference's shallow cv::Mat alias, stereo_slam.cpp:115): nothing is copied
            is->left[0] = ImgView{left[s], c->width, c->height, stride};
            is->right = ImgView{right[s], c->width, c->height, stride};
            hs->src_left = is->left[0];
        } else {
            // frames are ingested by the pyramid kernel itself (one launch for all sequences instead of
            // 2 copies per sequence); host-resident ones come through the staging buffer filled above
            is->left[0] = is->own_left0;
            is->right = is->own_right;
            const uint8_t* src_l = mem == SVO_MEM_DEVICE ? left[s] : c->d_stage_in + (size_t)s * c->stage_frame_bytes;
            const uint8_t* src_r = mem == SVO_MEM_DEVICE ? right[s] : c->d_stage_in + (size_t)(c->B + s) * c->stage_frame_bytes;
            hs->src_left = ImgView{src_l, c->width, c->height, stride};
            hs->src_right = ImgView{src_r, c->width, c->height, stride};
            hs->dst_right = is->right;
        }
        is->lk[0] = is->left[0];
        for (int l = 0; l < hs->n_levels; l++) hs->level[l] = is->left[l];
        hs->n_lk = c->n_lk;
        for (int l = 0; l < c->n_lk; l++) hs->lk[l] = is->lk[l];
        {
            const int rows = pyr_stream_rows(*hs);
            pyr_stream = (pyr_stream == 0 || rows == 0) ? 0 : std::max(pyr_stream, rows);
        }
    }

    if (!first) {
        auto fill = [c](int slot, int s) {
            Seq& q = c->seqs[s];
            FrameResult* dr = c->d_res + s;
            // predicted pose = kf.statePre (stereo_slam.cpp:183-192)
            float* guess = args_at<float>(c, c->off_guess, s * 8);
            // (== statePost while the previous frame's filter update is still pending, dt = 0)
            for (int i = 0; i < 6; i++) guess[i] = q.pending ? q.kf.statePost[i] : q.kf.statePre[i];
            const float* d_guess = dargs_at<float>(c, c->off_guess, s * 8);
            // remove_outliers: previous set -> other set (becomes the frame's keypoints)
            CompactArgs* ca = args_at<CompactArgs>(c, c->off_compact, slot);
            std::memset(ca, 0, sizeof(*ca));
            ca->src = q.kps[q.cur]; ca->dst = q.kps[q.cur ^ 1]; ca->mode = 0;
            ca->min_kf = &dr->min_kf;
            q.cur ^= 1;
            const KpsDev& k = q.kps[q.cur];
            SiaArgs* sa = args_at<SiaArgs>(c, c->off_sia, slot);
            std::memset(sa, 0, sizeof(*sa));
            for (int l = 0; l < c->cam.max_pyramid_levels; l++) {
                sa->prev[l] = q.prev_set->left[l];
                sa->cur[l] = q.cur_set->left[l];
            }
            sa->cam = c->cam; sa->n_ptr = k.n; sa->kps2d = k.kps2d; sa->kps3d = k.kps3d; sa->flags = k.flags;
            sa->pose_guess = d_guess; sa->pose_out = dr->pose_sia; sa->cost_out = &dr->sia_cost;
            sa->trace = dr->sia_trace; sa->kp_ws = q.sia_kpws;
            sa->rec_ws = q.sia_rec; sa->rec_cap = c->rec_cap;
            sa->mats_out = q.sia_mats;
            sa->dbg_H = nullptr; sa->dbg_level = -1; sa->cap = c->cap; sa->exact_pinv = c->exact_pinv;
            KltArgs* ka = args_at<KltArgs>(c, c->off_klt, slot);
            std::memset(ka, 0, sizeof(*ka));
            ka->kfs = q.d_kfs; ka->kf_id = k.kf_id; ka->n_cur = c->n_lk;
            for (int l = 0; l < c->n_lk; l++) ka->cur[l] = q.cur_set->lk[l];
            ka->n_ptr = k.n; ka->prev_pts = nullptr; ka->cur_pts = q.tracked; ka->status = q.klt_status;
            ka->err = q.klt_err; ka->win = c->cam.window_size_opt_flow;
            ka->proj_pose = dr->pose_sia; ka->proj_mats = q.sia_mats; ka->kps3d = k.kps3d; ka->proj_out = k.kps2d;
            ka->kp_index = k.kp_index; ka->ref_out = nullptr; ka->cam = c->cam;
            ReprojArgs* ra = args_at<ReprojArgs>(c, c->off_rp, slot);
            std::memset(ra, 0, sizeof(*ra));
            ra->cam = c->cam; ra->n_ptr = k.n; ra->kps2d = k.kps2d; ra->kps3d = k.kps3d; ra->flags = k.flags;
            ra->tracked = q.tracked; ra->err = q.klt_err; ra->pose_in = dr->pose_sia;
            ra->pose_out = dr->pose_refined; ra->cost_out = &dr->reproj_cost; ra->trace = &dr->reproj_trace;
            ra->exact_pinv = c->exact_pinv;
            ra->zero_out = c->d_inside + s;      // filter_update_kernel adds to it
            SsdArgs* ss = args_at<SsdArgs>(c, c->off_ssd, slot);
            std::memset(ss, 0, sizeof(*ss));
            ss->left = q.cur_set->left[0]; ss->right = q.cur_set->right; ss->n_ptr = k.n;
            ss->kps2d = k.kps2d; ss->disparity = q.disparity;
            ss->win = c->cam.window_size_depth_calculator; ss->search_x = c->cam.search_x;
            ss->search_y = c->cam.search_y; ss->clamp_half = 1;
            FilterArgs* fa = args_at<FilterArgs>(c, c->off_filt, slot);
            std::memset(fa, 0, sizeof(*fa));
            fa->cam = c->cam; fa->n_ptr = k.n; fa->frame_pose = dr->pose_refined;
            fa->kps2d = k.kps2d; fa->kps3d = k.kps3d; fa->flags = k.flags;
            fa->outlier_count = k.outl; fa->inlier_count = k.inl; fa->kf_inv_depth = k.kfx;
            fa->kf_variance = k.kfP; fa->disparity = q.disparity;
            fa->kfs = q.d_kfs; fa->kf_id = k.kf_id; fa->kp_index = k.kp_index;
            fa->do_outlier_check = 1; fa->do_update = 1; fa->do_flags = 1; fa->do_reproject = 1;
            fa->width = c->width; fa->height = c->height; fa->inside_count = c->d_inside + s;
        };
        for (int j = 0; j < M; j++) fill(j, act[j]);   // ~20 us for 256 sequences: not worth waking the pool
    }
    hlap(0);   // argument blocks
    HIP_TRY(hipMemcpyAsync(c->d_args, c->h_args, first ? c->args_bytes : c->frame_args_bytes,
                           hipMemcpyHostToDevice, c->stream));
    launch_pyr_fused(dargs_at<PyrArgs>(c, c->off_hs), M, c->width, c->height, mem != SVO_MEM_DEVICE_BORROW, std::max(pyr_stream, 0), c->stream);
    HIP_TRY(hipGetLastError());   // (every launch is checked on its own: a later success must not mask a failure)
    std::vector<int> need(B, 0);
    if (first) {
        for (int s = 0; s < B; s++) {
            need[s] = 1;
            HIP_TRY(hipMemsetAsync(c->seqs[s].d_n, 0, sizeof(int) * 2, c->stream));
        }
        HIP_TRY(hipMemsetAsync(c->d_res, 0, sizeof(FrameResult) * B, c->stream));
        if ((rc = enqueue_keyframes(c, need, true))) return rc;
    } else {
        SVO_MARK(1);
        launch_compact(dargs_at<CompactArgs>(c, c->off_compact), M, c->cap, c->stream);
        HIP_TRY(hipGetLastError());
        SVO_MARK(2);
        // the compaction can only shrink a sequence's keypoint set, so last frame's counts bound the
        // grids and the alignment kernel's LDS working set
        int grid_n = 1;
        for (int j = 0; j < M; j++) grid_n = std::max(grid_n, c->seqs[act[j]].n_host);
        grid_n = std::min(grid_n, c->cap);
        if (!launch_sia(dargs_at<SiaArgs>(c, c->off_sia), M, c->cam, c->width, c->height, grid_n, c->rec_cap, c->exact_pinv, c->stream)) {
            HIP_TRY(hipGetLastError());      // (the LDS limit of the kernel could not be raised on this device)
            return svo_set_error(SVO_ERR_CAPACITY, "sparse alignment: %d keypoints exceed the workspaces", grid_n);
        }
        HIP_TRY(hipGetLastError());
        SVO_MARK(3);
        launch_klt(dargs_at<KltArgs>(c, c->off_klt), M, grid_n, c->cam.window_size_opt_flow, c->stream);
        HIP_TRY(hipGetLastError());
        SVO_MARK(4);
        if (!launch_reproj(dargs_at<ReprojArgs>(c, c->off_rp), M, grid_n, c->stream)) {
            HIP_TRY(hipGetLastError());
            return svo_set_error(SVO_ERR_CAPACITY, "reprojection GN: %d keypoints do not fit LDS", grid_n);
        }
        HIP_TRY(hipGetLastError());
        SVO_MARK(5);
        launch_ssd(dargs_at<SsdArgs>(c, c->off_ssd), M, grid_n, c->cam.window_size_depth_calculator, c->cam.search_y, c->stream);
        HIP_TRY(hipGetLastError());
        SVO_MARK(6);
        launch_filter(dargs_at<FilterArgs>(c, c->off_filt), M, grid_n, c->stream);
        HIP_TRY(hipGetLastError());
        SVO_MARK(7);
        HIP_TRY(hipMemcpyAsync(c->h_inside, c->d_inside, sizeof(int) * B, hipMemcpyDeviceToHost, c->stream));
        hlap(1);   // launches
        flush_pending(c);                 // previous frame's pose filter, overlapped with the kernels
        hlap(2);   // pose filter
        HIP_TRY(hipStreamSynchronize(c->stream));
        hlap(3);   // wait for the frame
        // KeyFrameManager::keyframe_needed (keyframe_manager.cpp:66-72)
        const int max_keypoints = (c->width / c->cam.grid_width) * (c->height / c->cam.grid_height);
        bool any = false;
        for (int j = 0; j < M; j++) {
            const int s = act[j];
            need[s] = (double)c->h_inside[s] < 0.66 * max_keypoints ? 1 : 0;
            any = any || need[s];
        }
        if (any && (rc = enqueue_keyframes(c, need, false))) return rc;
        hlap(4);   // keyframe enqueue
    }
    HIP_TRY(hipMemcpyAsync(c->h_res, c->d_res, c->readback_bytes, hipMemcpyDeviceToHost, c->stream));   // results + counts
    HIP_TRY(hipStreamSynchronize(c->stream));
    hlap(5);   // wait for keyframes + read-back

    SVO_MARK(8);
    float stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c->timing) {
        HIP_TRY(hipEventSynchronize(c->ev[8]));
        if (!first) {
            for (int i = 0; i < 7; i++) (void)hipEventElapsedTime(&stage_ms[i], c->ev[i], c->ev[i + 1]);
            (void)hipEventElapsedTime(&stage_ms[7], c->ev[7], c->ev[8]);
        } else {
            (void)hipEventElapsedTime(&stage_ms[7], c->ev[0], c->ev[8]);
        }
    }
    const float sia_ms = stage_ms[2];

    // ---- host bookkeeping (stereo_slam.cpp:250-270); the pose filter itself is deferred
    int overflow_seq = -1;
    for (int j = 0; j < M; j++) {
        const int s = act[j];
        Seq& q = c->seqs[s];
        const FrameResult& r = c->h_res[s];
        const double ts = (double)time_stamps[s];
        q.frame_id++;
        if (first) {
            std::memset(q.pose, 0, sizeof(q.pose));
            q.ts = ts;
            svo_pose p;
            std::memcpy(&p, q.pose, sizeof(p));
            q.trajectory.push_back(p);
        } else {
            q.pending = true;
            std::memcpy(q.pending_pose, r.pose_refined, sizeof(q.pending_pose));
            q.pending_ts = ts;
        }
        if (need[s]) {
            KfHost& k = q.kfs.back();
            k.n = r.kf_n;
            if (first) std::memset(k.pose, 0, sizeof(k.pose));
            else std::memcpy(k.pose, r.pose_refined, sizeof(k.pose));
        }
        // Keyframe images are only read for keypoints that came from that keyframe (KLT builds a template from
        // them when the cache has none). The frame's keypoints — kept by the compaction at its start, plus what a
        // keyframe created in this frame adds — refer to keyframes r.min_kf and younger and, of the next 64, to
        // those whose bit is set in r.live_kf: the others hand their image sets back to the sequence's free
        // list, so memory stays bounded by the keyframes still in use
        // instead of growing with every keyframe (the reference keeps them all). Nothing else of a keyframe goes:
        // its keypoint arrays, pose and table record stay for the depth filter and the getters.
        if (!first && c->retire_kf_images) {
            const int newest = (int)q.kfs.size() - 1;                  // (never the newest: a keyframe made in this frame)
            const int upto = std::min(r.min_kf, newest);
            for (; q.kfs_retired < upto; q.kfs_retired++) {
                KfHost& old = q.kfs[q.kfs_retired];
                release_set(q, old.set);
                old.set = nullptr;
            }
            for (int j = 0; j < 64 && r.min_kf < newest && r.min_kf + j < newest; j++) {
                KfHost& old = q.kfs[r.min_kf + j];
                if (old.set && !((r.live_kf[j >> 5] >> (j & 31)) & 1u)) {
                    release_set(q, old.set);
                    old.set = nullptr;
                }
            }
        }
        q.n_host = c->h_n[2 * s + q.cur];
        svo_frame_stats& st = q.stats;
        std::memset(&st, 0, sizeof(st));
        st.frame_id = q.frame_id; st.is_keyframe = need[s]; st.n_keypoints = q.n_host;
        st.n_keyframes = (int)q.kfs.size(); st.inside_count = first ? 0 : c->h_inside[s]; st.overflow = r.overflow;
        std::memcpy(st.pose_sia, r.pose_sia, sizeof(st.pose_sia));
        std::memcpy(st.pose_refined, r.pose_refined, sizeof(st.pose_refined));
        st.sia_cost = r.sia_cost; st.reproj_cost = r.reproj_cost; st.sia_ms = sia_ms;
        std::memcpy(st.stage_ms, stage_ms, sizeof(stage_ms));
        std::memcpy(st.sia_trace, r.sia_trace, sizeof(st.sia_trace));
        st.reproj_trace = r.reproj_trace;
        c->totals.frames++;
        c->totals.keyframes += need[s];
        c->totals.keypoints += q.n_host;
        if (!first)
            for (int l = 0; l < SVO_MAX_PYRAMID_LEVELS; l++) {
                c->totals.gn_gradient_calls += r.sia_trace[l].n_gradient;
                c->totals.gn_cost_calls += r.sia_trace[l].n_cost;
            }
        if (r.overflow && overflow_seq < 0) overflow_seq = s;     // reported after every sequence is booked
    }
    hlap(6);   // bookkeeping
    c->host_steps++;
    c->totals.launches++;
    for (int i = 0; i < 8; i++) c->totals.stage_ms[i] += stage_ms[i];
    c->totals.wall_ms +=
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (overflow_seq >= 0)
        return svo_set_error(SVO_ERR_CAPACITY, "sequence %d: more than %d keypoints", overflow_seq, c->cap);
    return SVO_OK;
}

// A frame that fails half way (HIP error, capacity) leaves the sequences of the group at mixed
// frame ids: the group is marked failed and rejects further frames instead of tracking on.
static int grp_new_images(svo_group* c, const uint8_t* const* left, const uint8_t* const* right,
                          int stride, const float* time_stamps, int mem) {
    if (!c || !left || !right || !time_stamps || stride < c->width)
        return svo_set_error(SVO_ERR_INVALID, "svo_new_images: bad arguments");
    if (c->failed)
        return svo_set_error(SVO_ERR_INVALID, "svo_new_images: an earlier frame of this ctx failed; create a new ctx");
    const int rc = grp_new_images_impl(c, left, right, stride, time_stamps, mem);
    if (rc != SVO_OK) c->failed = true;
    return rc;
}

static int grp_get_totals(svo_group* c, svo_totals* out) {
    if (!c || !out) return svo_set_error(SVO_ERR_INVALID, "svo_get_totals: bad arguments");
    *out = c->totals;
    out->image_sets = c->image_sets;
    return SVO_OK;
}

static int grp_new_image(svo_group* c, const uint8_t* left, int left_stride, const uint8_t* right,
                             int right_stride, int width, int height, float time_stamp) {
    if (!c || c->B != 1) return svo_set_error(SVO_ERR_INVALID, "svo_new_image needs a 1-sequence ctx");
    if (width != c->width || height != c->height || left_stride != right_stride)
        return svo_set_error(SVO_ERR_INVALID, "svo_new_image: image size / stride mismatch");
    return grp_new_images(c, &left, &right, left_stride, &time_stamp, SVO_MEM_HOST);
}

#define CHECK_SEQ(c, seq)                                                              \
    do {                                                                               \
        if (!(c) || (seq) < 0 || (seq) >= (c)->B)                                      \
            return svo_set_error(SVO_ERR_INVALID, "bad ctx / sequence index");         \
        HIP_TRY(hipSetDevice((c)->device));                                            \
    } while (0)

static int grp_get_pose(svo_group* c, int seq, float pose[6]) {
    CHECK_SEQ(c, seq);
    flush_pending(c);
    std::memcpy(pose, c->seqs[seq].pose, sizeof(float) * 6);
    return SVO_OK;
}

static int fetch_info(svo_group* c, int n, const svo_kp2d* d2, const svo_kp3d* d3, const uint32_t* dfl,
                      const int* dkf, const int* dki, const int* dout, const int* din, const float* dkx,
                      const float* dkP, const float* dsc, const int* dlt, const uint32_t* dcol,
                      svo_kp2d* kps2d, svo_kp3d* kps3d, svo_kp_info* info) {
    if (n <= 0) return SVO_OK;
    if (kps2d) HIP_TRY(hipMemcpy(kps2d, d2, sizeof(svo_kp2d) * n, hipMemcpyDeviceToHost));
    if (kps3d) HIP_TRY(hipMemcpy(kps3d, d3, sizeof(svo_kp3d) * n, hipMemcpyDeviceToHost));
    if (!info) return SVO_OK;
    std::vector<uint32_t> fl(n), col(n, 0);
    std::vector<int> kf(n, 0), ki(n, 0), ou(n), in(n), lt(n, 0);
    std::vector<float> kx(n, 0), kP(n, 0), sc(n, 0);
    HIP_TRY(hipMemcpy(fl.data(), dfl, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ou.data(), dout, sizeof(int) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(in.data(), din, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (dkf) HIP_TRY(hipMemcpy(kf.data(), dkf, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (dki) HIP_TRY(hipMemcpy(ki.data(), dki, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (dkx) HIP_TRY(hipMemcpy(kx.data(), dkx, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (dkP) HIP_TRY(hipMemcpy(kP.data(), dkP, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (dsc) HIP_TRY(hipMemcpy(sc.data(), dsc, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (dlt) HIP_TRY(hipMemcpy(lt.data(), dlt, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (dcol) HIP_TRY(hipMemcpy(col.data(), dcol, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        svo_kp_info& o = info[i];
        std::memset(&o, 0, sizeof(o));
        o.score = sc[i]; o.level = lt[i] & 0xff; o.type = (lt[i] >> 8) & 0xff;
        o.keyframe_id = kf[i]; o.keypoint_index = ki[i];
        o.color[0] = col[i] & 0xff; o.color[1] = (col[i] >> 8) & 0xff; o.color[2] = (col[i] >> 16) & 0xff;
        o.ignore_during_refinement = (fl[i] & SVO_IGNORE_DURING_REFINEMENT) != 0;
        o.ignore_completely = (fl[i] & SVO_IGNORE_COMPLETELY) != 0;
        o.ignore_temporary = (fl[i] & SVO_IGNORE_TEMPORARY) != 0;
        o.outlier_count = ou[i]; o.inlier_count = in[i];
        o.kf_inv_depth = kx[i]; o.kf_variance = kP[i];
    }
    return SVO_OK;
}

static int grp_get_frame_keypoints(svo_group* c, int seq, svo_kp2d* kps2d, svo_kp3d* kps3d,
                                       svo_kp_info* info, int cap, int* n) {
    CHECK_SEQ(c, seq);
    Seq& q = c->seqs[seq];
    if (n) *n = q.n_host;
    const KpsDev& k = q.kps[q.cur];
    return fetch_info(c, std::min(cap, q.n_host), k.kps2d, k.kps3d, k.flags, k.kf_id, k.kp_index,
                      k.outl, k.inl, k.kfx, k.kfP, k.score, k.level_type, k.color, kps2d, kps3d, info);
}

static int grp_get_keyframe_count(svo_group* c, int seq, int* count) {
    CHECK_SEQ(c, seq);
    if (count) *count = (int)c->seqs[seq].kfs.size();
    return SVO_OK;
}

static int grp_get_keyframe(svo_group* c, int seq, int id, svo_kp2d* kps2d, svo_kp3d* kps3d,
                                svo_kp_info* info, float pose[6], int cap, int* n) {
    CHECK_SEQ(c, seq);
    Seq& q = c->seqs[seq];
    if (id < 0 || id >= (int)q.kfs.size()) return svo_set_error(SVO_ERR_INVALID, "keyframe %d does not exist", id);
    const KfHost& k = q.kfs[id];
    if (n) *n = k.n;
    if (pose) std::memcpy(pose, k.pose, sizeof(float) * 6);
    return fetch_info(c, std::min(cap, k.n), k.kps2d, k.kps3d, k.flags, k.kf_id, k.kp_index, k.outl, k.inl,
                      k.kfx, k.kfP, k.score, k.level_type, k.color, kps2d, kps3d, info);
}

static int grp_get_trajectory(svo_group* c, int seq, svo_pose* out, int cap, int* n) {
    CHECK_SEQ(c, seq);
    flush_pending(c);
    Seq& q = c->seqs[seq];
    if (n) *n = (int)q.trajectory.size();
    const int m = std::min<int>(cap, (int)q.trajectory.size());
    if (out && m > 0) std::memcpy(out, q.trajectory.data(), sizeof(svo_pose) * m);
    return SVO_OK;
}

static int grp_update_pose(svo_group* c, int seq, const float pose[6], const float speed[6],
                               const float pose_var[6], const float speed_var[6], double dt,
                               float filtered[6]) {
    CHECK_SEQ(c, seq);
    flush_pending(c);
    c->seqs[seq].kf.update(pose, speed, pose_var, speed_var, dt, filtered);
    return SVO_OK;
}

static int grp_get_frame_stats(svo_group* c, int seq, svo_frame_stats* out) {
    CHECK_SEQ(c, seq);
    if (out) *out = c->seqs[seq].stats;
    return SVO_OK;
}

// =====================================================================================
// svo_ctx: the public object. Its sequences are split over 1..G groups; a group owns a
// HIP stream, its argument blocks and (G > 1) a host thread that drives it, so the groups
// run their frames independently: while one group waits for its keyframe decision or fills
// its argument blocks, the other group's kernels keep the GPU busy, and the latency-bound
// single-workgroup-per-sequence alignment kernel of one group overlaps the window kernels of
// the other. svo_submit_images() queues a frame set on every group and returns;
// svo_wait() drains the queues. svo_new_images() = submit + wait.
// =====================================================================================
struct svo_ctx {
    struct Job {
        std::vector<const uint8_t*> left, right;
        std::vector<float> ts;
        int stride, mem;
    };
    struct Worker {
        svo_group* g = nullptr;
        int first = 0, count = 0;
        std::thread th;
        std::mutex m;
        std::condition_variable cv, cv_idle;
        std::deque<Job> jobs;
        bool busy = false, stop = false;
        int err = SVO_OK;
        std::string msg;
        std::atomic<bool>* ctx_failed = nullptr;
    };
    int B = 0, device = 0;
    // A frame that fails in ONE group leaves the ctx's sequences at mixed frame ids: the failure is
    // latched here, the other groups drop what is still queued, and later submits are rejected.
    std::atomic<bool> failed{false};
    std::vector<std::unique_ptr<Worker>> workers;
};

namespace {

void worker_run_job(svo_ctx::Worker& w, const svo_ctx::Job& job) {
    if (w.err != SVO_OK || w.ctx_failed->load()) return;   // after a failure (any group) the queues are dropped
    const int rc = grp_new_images(w.g, job.left.data(), job.right.data(), job.stride, job.ts.data(), job.mem);
    if (rc != SVO_OK) {
        w.err = rc;
        w.msg = svo_last_error();
        w.ctx_failed->store(true);
    }
}

void worker_loop(svo_ctx::Worker* w) {
    for (;;) {
        svo_ctx::Job job;
        {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [w] { return w->stop || !w->jobs.empty(); });
            if (w->jobs.empty()) return;         // stop requested and nothing left
            job = std::move(w->jobs.front());
            w->jobs.pop_front();
            w->busy = true;
        }
        worker_run_job(*w, job);
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->busy = false;
            if (w->jobs.empty()) w->cv_idle.notify_all();
        }
    }
}

// wait for every group's queue; returns the first stored error (and clears it)
int ctx_drain(svo_ctx* c) {
    int rc = SVO_OK;
    for (auto& wp : c->workers) {
        svo_ctx::Worker& w = *wp;
        if (w.th.joinable()) {
            std::unique_lock<std::mutex> lk(w.m);
            w.cv_idle.wait(lk, [&w] { return w.jobs.empty() && !w.busy; });
        }
        if (w.err != SVO_OK && rc == SVO_OK) {
            rc = svo_set_error(w.err, "%s", w.msg.c_str());
            w.err = SVO_OK;
        }
    }
    return rc;
}

svo_ctx::Worker* ctx_locate(svo_ctx* c, int seq, int* local) {
    for (auto& wp : c->workers)
        if (seq >= wp->first && seq < wp->first + wp->count) {
            *local = seq - wp->first;
            return wp.get();
        }
    return nullptr;
}

}  // namespace

#define CTX_SEQ(c, seq, w, local)                                                     \
    int local = 0;                                                                     \
    svo_ctx::Worker* w = nullptr;                                                      \
    do {                                                                               \
        if (!(c) || (seq) < 0 || (seq) >= (c)->B)                                      \
            return svo_set_error(SVO_ERR_INVALID, "bad ctx / sequence index");         \
        int rc_ = ctx_drain(c);                                                        \
        if (rc_) return rc_;                                                           \
        w = ctx_locate(c, seq, &local);                                                \
    } while (0)

extern "C" int svo_ctx_create(const svo_camera_settings* cam, int width, int height, int n_sequences,
                              int device, svo_ctx** out) {
    if (!cam || !out || n_sequences < 1) return svo_set_error(SVO_ERR_INVALID, "svo_ctx_create: bad arguments");
    // SVO_GROUPS: number of independently driven groups. Default: groups of ~256 sequences — fewer per group
    // when a sequence carries many keypoints (131072 / keypoint capacity, at least 32: 64 at the 1920x1080
    // configuration, whose alignment launch is as slow as its slowest sequence whatever the group's size:
    // 256 sequences, frames/s: 1 group 12.1 K, 2 13.2 K, 4 14.1 K, 8 13.2 K; 512 in 8 18.2 K) —, at least
    // two from 64 sequences on, and one fewer than the hardware queues the HIP runtime uses
    // (GPU_MAX_HW_QUEUES, default 4): streams beyond that share a queue and serialise. Measured on
    // MI355X, 752x480, frames/s: 768 sequences 149 K as 3 groups, 106 K as 4, 119 K as 6 with 4
    // queues; with GPU_MAX_HW_QUEUES=8: 768 / 3 groups 154 K, 1024 / 4 162 K, 1536 / 6 170 K.
    int hwq = 4;
    if (const char* e = std::getenv("GPU_MAX_HW_QUEUES")) hwq = std::max(2, std::atoi(e));
    const int kp_cap = (cam->grid_width > 0 && cam->grid_height > 0 ? (width / cam->grid_width) * (height / cam->grid_height) : 0) + 64;
    const int per_group = std::max(32, std::min(256, 131072 / std::max(kp_cap, 1)));
    int G = n_sequences >= 64 ? std::max(2, (n_sequences + per_group / 2) / per_group) : 1;
    G = std::min(G, hwq - 1);
    if (const char* e = std::getenv("SVO_GROUPS")) G = std::atoi(e);
    G = std::max(1, std::min(G, std::min(n_sequences, 16)));
    svo_ctx* c = new (std::nothrow) svo_ctx();
    if (!c) return svo_set_error(SVO_ERR_INVALID, "out of host memory");
    c->B = n_sequences; c->device = device;
    int first = 0;
    for (int g = 0; g < G; g++) {
        const int count = n_sequences / G + (g < n_sequences % G ? 1 : 0);
        auto w = std::make_unique<svo_ctx::Worker>();
        w->first = first; w->count = count; w->ctx_failed = &c->failed;
        const int rc = grp_create(cam, width, height, count, device, &w->g);
        if (rc) {
            for (auto& o : c->workers) grp_destroy(o->g);
            delete c;
            return rc;
        }
        first += count;
        c->workers.push_back(std::move(w));
    }
    if (G > 1)
        for (auto& w : c->workers) w->th = std::thread(worker_loop, w.get());
    *out = c;
    return SVO_OK;
}

extern "C" int svo_ctx_destroy(svo_ctx* c) {
    if (!c) return SVO_OK;
    (void)ctx_drain(c);
    for (auto& w : c->workers) {
        if (w->th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(w->m);
                w->stop = true;
            }
            w->cv.notify_all();
            w->th.join();
        }
        grp_destroy(w->g);
    }
    delete c;
    return SVO_OK;
}

extern "C" int svo_ctx_get_groups(svo_ctx* c, int* n_groups) {
    if (!c || !n_groups) return svo_set_error(SVO_ERR_INVALID, "svo_ctx_get_groups: bad arguments");
    *n_groups = (int)c->workers.size();
    return SVO_OK;
}

extern "C" int svo_submit_images(svo_ctx* c, const uint8_t* const* left, const uint8_t* const* right,
                                 int stride, const float* time_stamps, int mem) {
    if (!c || !left || !right || !time_stamps) return svo_set_error(SVO_ERR_INVALID, "svo_submit_images: bad arguments");
    if (c->failed.load()) {                      // nothing is queued on any group once one of them has failed
        const int rc = ctx_drain(c);             // (the first call after the failure reports its cause)
        return rc ? rc : svo_set_error(SVO_ERR_INVALID, "svo_submit_images: an earlier frame of this ctx failed; create a new ctx");
    }
    for (auto& wp : c->workers) {
        svo_ctx::Worker& w = *wp;
        svo_ctx::Job job;
        job.left.assign(left + w.first, left + w.first + w.count);
        job.right.assign(right + w.first, right + w.first + w.count);
        job.ts.assign(time_stamps + w.first, time_stamps + w.first + w.count);
        job.stride = stride; job.mem = mem;
        if (w.th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(w.m);
                w.jobs.push_back(std::move(job));
            }
            w.cv.notify_one();
        } else {
            worker_run_job(w, job);              // single group: runs on the caller's thread
        }
    }
    return SVO_OK;
}

extern "C" int svo_wait(svo_ctx* c) {
    if (!c) return svo_set_error(SVO_ERR_INVALID, "svo_wait: bad ctx");
    return ctx_drain(c);
}

extern "C" int svo_new_images(svo_ctx* c, const uint8_t* const* left, const uint8_t* const* right,
                              int stride, const float* time_stamps, int mem) {
    const int rc = svo_submit_images(c, left, right, stride, time_stamps, mem);
    return rc ? rc : svo_wait(c);
}

extern "C" int svo_new_image(svo_ctx* c, const uint8_t* left, int left_stride, const uint8_t* right,
                             int right_stride, int width, int height, float time_stamp) {
    if (!c || c->B != 1) return svo_set_error(SVO_ERR_INVALID, "svo_new_image needs a 1-sequence ctx");
    return grp_new_image(c->workers[0]->g, left, left_stride, right, right_stride, width, height, time_stamp);
}

extern "C" int svo_ctx_set_exact_pinv(svo_ctx* c, int on);
extern "C" int svo_ctx_set_fast_solver(svo_ctx* c, int on) { return svo_ctx_set_exact_pinv(c, on == 0); }

extern "C" int svo_ctx_set_exact_pinv(svo_ctx* c, int on) {
    if (!c) return svo_set_error(SVO_ERR_INVALID, "bad ctx");
    int rc = ctx_drain(c);
    for (auto& w : c->workers)
        if (!rc) rc = grp_set_exact_pinv(w->g, on);
    return rc;
}

extern "C" int svo_ctx_enable_timing(svo_ctx* c, int on) {
    if (!c) return svo_set_error(SVO_ERR_INVALID, "bad ctx");
    int rc = ctx_drain(c);
    for (auto& w : c->workers)
        if (!rc) rc = grp_enable_timing(w->g, on);
    return rc;
}

extern "C" int svo_get_totals(svo_ctx* c, svo_totals* out) {
    if (!c || !out) return svo_set_error(SVO_ERR_INVALID, "svo_get_totals: bad arguments");
    int rc = ctx_drain(c);
    if (rc) return rc;
    std::memset(out, 0, sizeof(*out));
    for (auto& w : c->workers) {
        svo_totals t;
        if ((rc = grp_get_totals(w->g, &t))) return rc;
        out->frames += t.frames; out->keyframes += t.keyframes; out->keypoints += t.keypoints;
        out->gn_gradient_calls += t.gn_gradient_calls; out->gn_cost_calls += t.gn_cost_calls;
        for (int i = 0; i < 8; i++) out->stage_ms[i] += t.stage_ms[i];
        out->wall_ms = std::max(out->wall_ms, t.wall_ms);
        out->launches += t.launches;
        out->image_sets += t.image_sets;
    }
    out->n_groups = (int)c->workers.size();
    return SVO_OK;
}

extern "C" int svo_get_pose(svo_ctx* c, int seq, float pose[6]) {
    CTX_SEQ(c, seq, w, local);
    return grp_get_pose(w->g, local, pose);
}
extern "C" int svo_get_frame_keypoints(svo_ctx* c, int seq, svo_kp2d* kps2d, svo_kp3d* kps3d,
                                       svo_kp_info* info, int cap, int* n) {
    CTX_SEQ(c, seq, w, local);
    return grp_get_frame_keypoints(w->g, local, kps2d, kps3d, info, cap, n);
}
extern "C" int svo_get_keyframe_count(svo_ctx* c, int seq, int* count) {
    CTX_SEQ(c, seq, w, local);
    return grp_get_keyframe_count(w->g, local, count);
}
extern "C" int svo_get_keyframe(svo_ctx* c, int seq, int id, svo_kp2d* kps2d, svo_kp3d* kps3d,
                                svo_kp_info* info, float pose[6], int cap, int* n) {
    CTX_SEQ(c, seq, w, local);
    return grp_get_keyframe(w->g, local, id, kps2d, kps3d, info, pose, cap, n);
}
extern "C" int svo_get_trajectory(svo_ctx* c, int seq, svo_pose* out, int cap, int* n) {
    CTX_SEQ(c, seq, w, local);
    return grp_get_trajectory(w->g, local, out, cap, n);
}
extern "C" int svo_update_pose(svo_ctx* c, int seq, const float pose[6], const float speed[6],
                               const float pose_var[6], const float speed_var[6], double dt,
                               float filtered[6]) {
    CTX_SEQ(c, seq, w, local);
    return grp_update_pose(w->g, local, pose, speed, pose_var, speed_var, dt, filtered);
}
extern "C" int svo_get_frame_stats(svo_ctx* c, int seq, svo_frame_stats* out) {
    CTX_SEQ(c, seq, w, local);
    return grp_get_frame_stats(w->g, local, out);
}
