// sia.hip — rows A1-A9 of SURVEY §8a: sparse image alignment,
// PoseEstimator::estimate_pose (src/lib/pose_estimator.cpp:115-130) with
// estimate_pose_at_level (:166-222), do_calc (:275-300), calculate_hessian
// (:312-416), get_gradient (:418-539), setLevel (:541-562) and the patch cost
// of src/lib/image_comparison.cpp:9-120.
//
// ONE persistent workgroup per sequence runs every pyramid level and every
// Gauss-Newton / line-search iteration on the device (the reference makes up
// to 50 cost evaluations per level; a launch per evaluation would cost more
// than the whole CPU frame):
//
//   sia_prep_kernel            everything that depends only on the PREVIOUS frame, per patch pixel:
//                              the reference half of the cost, the reference patch sum and the image
//                              gradient, sum g g^T per keypoint — once per frame, fully parallel;
//   sia_gn_kernel<WAVES, MODE> 64*WAVES lanes, one keypoint per lane and pass.
//
//   * The current level image (23x15 .. 188x120 bytes at 752x480) and the records are staged in
//     LDS once per level (MODE 0), only the cost records (MODE 1), or read from L2 (MODE 2);
//     a cost evaluation reads the 5x5 window of its patch, a gradient evaluation the 6x6 window.
//   * Everything that is the same for all keypoints — Rodrigues, the 6x6 solve, the exponential
//     map, the accept / halve / stop logic — is computed redundantly by every lane: no
//     broadcast, no barrier. With WAVES == 1 the kernel has no barrier at all.
//   * Float reductions follow the reference's order. The cost is summed per keypoint over its 16
//     pixels in the lane (image_comparison.cpp:67-88) and then over the keypoints in index order
//     (:112-117; v_readlane adds for one wave, an LDS pass otherwise): the stop test |dcost| < 1
//     is exact. The normal equations have two forms: the DEFAULT accumulates hessian += row^T row
//     and residual -= row * diff row by row in storage order (:399-403, :472-477) — 64 keypoints
//     stage their rows in LDS, 27 accumulator lanes walk them, four keypoints per trip (with four
//     waves: wave 0 only adds while the others compute and stage the next keypoints, two buffers) —
//     and inverts by the Jacobi SVD, so the whole iteration trace is the reference's; the fast solver
//     (svo_*_set_fast_solver) builds sum_kp J^T (sum_px g g^T) J with a wave reduction and solves
//     by LDL^T.
#include "svo_kernels.hpp"
#include <mutex>
#include <algorithm>
#include <cstdlib>

namespace svo {

constexpr size_t SIA_LDS_BUDGET = 156 * 1024;

#define LDSF(p) ((SVO_LDS(float)*)(p))
#define LDSCF(p) ((const SVO_LDS(float)*)(p))

// the current level image: the LDS copy (rows padded to a dword) or, for level images too large
// for LDS (BIG), the image in HBM / L2
template <bool BIG>
struct LevelImg {
    const uint8_t* p;
    int w, h, stride;
    __device__ inline float at(int o) const {
        if constexpr (BIG) return (float)((SVO_GP(const uint8_t))p)[o];
        else return (float)((const SVO_LDS(uint8_t)*)p)[o];
    }
};

// get_patch_sum, src/lib/pose_estimator.cpp:82-112, on the level image
template <bool BIG>
__device__ inline float patch_sum_lds(const LevelImg<BIG>& im, float cx, float cy) {
    const float sx = cx - 0.5f, sy = cy - 0.5f;
    const int ipx = (int)floorf(sx), ipy = (int)floorf(sy);
    const float x2 = sx - (float)ipx, y2 = sy - (float)ipy;
    const float x1 = 1.0f - x2, y1 = 1.0f - y2;
    const int o = ipy * im.stride + ipx;
    const float a00 = im.at(o), a01 = im.at(o + 1), a02 = im.at(o + 2);
    const float a10 = im.at(o + im.stride), a11 = im.at(o + im.stride + 1), a12 = im.at(o + im.stride + 2);
    const float a20 = im.at(o + 2 * im.stride), a21 = im.at(o + 2 * im.stride + 1),
                a22 = im.at(o + 2 * im.stride + 2);
    const float intensity = x1 * y1 * a00 + y1 * a01 + x2 * y1 * a02 +
                            x1 * a10 + a11 + x2 * a12 +
                            x1 * y2 * a20 + y2 * a21 + x2 * y2 * a22;
    return intensity;
}

// Dynamic LDS of one workgroup (byte offsets). cap = keypoint capacity (multiple of 64),
// T = threads. Per keypoint: 9 floats (point, last projection, sum g g^T, active) and the
// 64 per-pixel records, all struct-of-arrays with the keypoint index fastest (conflict free).
#ifndef SVO_SIA_STG
#define SVO_SIA_STG 64
#endif
// keypoints whose rows are staged at a time in reference-order mode: 64; the one-wave shape can be built with 32
// (-DSVO_SIA_STG=32: 18 instead of 36 KB of LDS per sequence, two staging rounds per pass)
__host__ __device__ constexpr int sia_stg(int T) { return T == 64 ? SVO_SIA_STG : 64; }
#ifndef SVO_SIA_ACC_U
#define SVO_SIA_ACC_U 4
#endif
constexpr int SIA_ACC_U = SVO_SIA_ACC_U;   // keypoints per trip of the ordered accumulation (their LDS reads are in flight together)    // keypoints whose rows are staged at a time in reference-order mode
enum { KF_PX = 0, KF_PY, KF_PZ, KF_QX, KF_QY, KF_GXX, KF_GXY, KF_GYY, KF_ACT, KF_COUNT };
enum { REC_I1 = 0, REC_PS = 1, REC_G0 = 2, REC_G1 = 3 };
struct SiaLds {
    size_t img, tbuf, kpf, rec, sums, stage, pipe, total;
};
__host__ __device__ inline SiaLds sia_lds_layout(int img_bytes, int cap, int T, bool exact, int mode = 0) {
    SiaLds l;
    size_t off = 0;
    l.img = off;   off += mode == 2 ? 0 : ((size_t)img_bytes + 15) & ~(size_t)15;
    l.tbuf = off;  off += T == 64 ? 0 : (size_t)2 * cap * 4;             // one wave sums by v_readlane
    l.kpf = off;   off += mode == 2 ? 0 : (size_t)KF_COUNT * cap * 4;
    l.rec = off;   off += mode == 2 ? 0 : (size_t)(mode == 1 ? 16 : 64) * cap * 4;
    l.sums = off;  off += (size_t)(T / 64) * 32 * 4 + 32;              // [WAVES][32] + the step wave 0 hands to the others
    // 7 planes of SIA_STG keypoints' rows (20 floats each); several waves per sequence fill two of them in turn
    l.stage = off; off += exact ? (size_t)(T > 128 ? 2 : 1) * 7 * (sia_stg(T) * 20 + 4) * 4 : 0;
    l.pipe = off;  off += exact && T > 128 ? 16 : 0;                   // hand-over counters of the two buffers
    l.total = off;
    return l;
}

template <int WAVES>
__device__ inline void sia_sync() {
    if constexpr (WAVES > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
}

// (r, c) of the 21 upper-triangle entries of H in the order they are stored
__constant__ int8_t c_tri_r[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
__constant__ int8_t c_tri_c[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};

// ---------------------------------------------------------------- records
// Everything the alignment needs from the PREVIOUS frame is independent of the pose: per patch
// pixel the reference half of the cost (image_comparison.cpp:20-88), the reference patch sum of
// the residual (:449-460) and the image gradient of calculate_hessian (:351-388), per keypoint
// sum g g^T. sia_prep_kernel computes them for every level in one launch, one lane per
// (keypoint, patch pixel), into a struct-of-arrays workspace: rec_ws[level - min][row][rec_cap]
// with row = field * 16 + pixel (fields REC_*), rows 64..66 = sum gxgx, gxgy, gygy.
constexpr int SIA_REC_ROWS = 68;

__global__ __launch_bounds__(64) void sia_prep_kernel(const SiaArgs* __restrict__ args) {
    const SiaArgs& a = args[blockIdx.z];
    const int n = min(*G(a.n_ptr), a.rec_cap);
    const int level = a.cam.min_pyramid_level_pose_estimation + blockIdx.y;
    if (level >= a.cam.max_pyramid_levels) return;
    const int idx = blockIdx.x * 64 + threadIdx.x;      // single-wave workgroups: they fit any free slot
    const int kp = idx >> 4, px = idx & 15;
    const int span = (n + 3) & ~3;                       // the consumer copies float4 columns
    if (kp >= span) return;                              // (whole 16-lane rows leave together)
    const ImgView prev = a.prev[level];
    const int divider = 1 << level;
    float g0 = 0, g1 = 0, psr = __builtin_nanf(""), i1 = __builtin_nanf("");
    const bool active = kp < n && !(a.flags && (G(a.flags)[kp] & SVO_IGNORE_TEMPORARY));
    if (active) {
        svo_kp2d kref = G(a.kps2d)[kp];
        if (level != 0) { kref.x /= divider; kref.y /= divider; }      // setLevel
        // the reference's walk over the patch: x++ per column, x -= 4 and y++ per row (float arithmetic)
        float kx = kref.x - 2.f, ky = kref.y - 2.f;
        for (int rr = 0; rr < (px >> 2); rr++) {
            kx += 1.f; kx += 1.f; kx += 1.f; kx += 1.f;
            kx -= 4.f;
            ky += 1.f;
        }
        for (int cc = 0; cc < (px & 3); cc++) kx += 1.f;
        // calculate_hessian bounds (:351-352)
        if (!(((double)kx - 2.0) < 0 || ((double)ky - 2.0) < 0 ||
              ((double)kx + 3.0) >= prev.w || ((double)ky + 3.0) >= prev.h)) {
            const float int1 = patch_sum(prev.g(), prev.stride, kx + 1, ky);
            const float int2 = patch_sum(prev.g(), prev.stride, kx - 1, ky);
            const float int3 = patch_sum(prev.g(), prev.stride, kx, ky + 1);
            const float int4 = patch_sum(prev.g(), prev.stride, kx, ky - 1);
            g0 = int1 - int2; g1 = int3 - int4;
        }
        // reference half of the residual test (:449-453)
        if (!(((double)kx - 1.0) < 0 || ((double)ky - 1.0) < 0 ||
              ((double)kx + 2.0) > prev.w || ((double)ky + 2.0) > prev.h))
            psr = patch_sum(prev.g(), prev.stride, kx, ky);
        // reference half of the cost (image_comparison.cpp:20-88): one window test per keypoint
        const int ps = a.cam.window_size_pose_estimator;
        const float half_size = ((float)ps - 1.0f) / 2.0f;
        const float s1x = kref.x - half_size, s1y = kref.y - half_size;
        const float f1x = floorf(s1x), f1y = floorf(s1y);
        if (f1x >= 0.f && f1y >= 0.f && f1x < 65536.f && f1y < 65536.f) {
            const int ip1x = (int)f1x, ip1y = (int)f1y;
            if (ip1y + ps < prev.h && ip1x + ps < prev.w) {
                const float x12 = s1x - (float)ip1x, y12 = s1y - (float)ip1y;
                const float x11 = 1.0f - x12, y11 = 1.0f - y12;
                const float m0 = x11 * y11, m1 = x12 * y11, m2 = x11 * y12, m3 = x12 * y12;
                const uint8_t* p = prev.g() + (size_t)((px >> 2) + ip1y) * prev.stride + (px & 3) + ip1x;
                float t = 0;
                t += m0 * (float)p[0];
                t += m1 * (float)p[1];
                t += m2 * (float)p[prev.stride];
                t += m3 * (float)p[prev.stride + 1];
                i1 = t;
            }
        }
    }
    float* out = G(a.rec_ws) + (size_t)blockIdx.y * SIA_REC_ROWS * a.rec_cap;
    out[(size_t)(REC_I1 * 16 + px) * a.rec_cap + kp] = i1;
    out[(size_t)(REC_PS * 16 + px) * a.rec_cap + kp] = psr;
    out[(size_t)(REC_G0 * 16 + px) * a.rec_cap + kp] = g0;
    out[(size_t)(REC_G1 * 16 + px) * a.rec_cap + kp] = g1;
    const float gxx = row16_sum_dpp(g0 * g0), gxy = row16_sum_dpp(g0 * g1), gyy = row16_sum_dpp(g1 * g1);
    if (px == 0) {
        out[(size_t)64 * a.rec_cap + kp] = gxx;
        out[(size_t)65 * a.rec_cap + kp] = gxy;
        out[(size_t)66 * a.rec_cap + kp] = gyy;
    }
}

// diagnostic build (-DSVO_SIA_STAMPS, tools/sia_stamps.py): cycles per phase, written to dbg_H
#ifdef SVO_SIA_STAMPS
#define SIA_T(v) const long long v = __builtin_readcyclecounter()
#define SIA_ADD(i, t1, t0) st[i] += (t1) - (t0)
#else
#define SIA_T(v)
#define SIA_ADD(i, t1, t0)
#endif

// MODE — where the working set of a sequence lives (same arithmetic, same order in all three):
//   0 (a few sequences, latency matters): level image, per-keypoint values and all records in LDS
//     (~100 KB);
//   1 (the step between the two, when MODE 0 does not fit): image, per-keypoint values and the cost
//     records (i1) in LDS, the records only get_gradient needs (reference patch sums, image
//     gradients, sum g g^T: 3/4 of the bytes) are read where sia_prep_kernel wrote them;
//   2 (a batch of sequences — LDS is what the window kernels of the other sequence groups run
//     short of — and keypoint sets / level images that do not fit LDS, the 1920x1080 configuration
//     with ~1700 keypoints and a 480x270 finest level): per-keypoint values in the HBM workspace
//     SiaArgs::kp_ws, records and image taps from L2; LDS holds only the staging area of the
//     ordered accumulation (~38 KB).
template <int WAVES, int MODE>
struct Sia {
    static constexpr bool BIG = MODE == 2;
    static constexpr int T = 64 * WAVES;
#ifdef SVO_SIA_STAMPS
    long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    const SiaArgs& a;
    int n, cap;
    int npad;                        // n rounded up to whole passes of the workgroup: the keypoint loops of THIS sequence
    uint8_t* dyn;
    SiaLds lay;
    LevelImg<BIG> cur;
    const float* recs = nullptr;     // BIG: records of the level in HBM, rows of a.rec_cap
    float fx, fy, cx, cy;
    int patch;
    int par = 0;

    __device__ Sia(const SiaArgs& a_, int n_, int cap_, uint8_t* dyn_, const SiaLds& lay_)
        : a(a_), n(n_), cap(cap_), npad(min(cap_, (n_ + T - 1) / T * T)), dyn(dyn_), lay(lay_) {}

    __device__ inline float kpf_ld(int f, int i) const {
        if constexpr (BIG) return G(a.kp_ws)[(size_t)f * cap + i];
        else return (LDSF(dyn + lay.kpf) + f * cap)[i];
    }
    __device__ inline void kpf_st(int f, int i, float v) const {
        if constexpr (BIG) G(a.kp_ws)[(size_t)f * cap + i] = v;
        else (LDSF(dyn + lay.kpf) + f * cap)[i] = v;
    }
    __device__ inline float rec_ld(int f, int px, int i) const {
        if (BIG || (MODE == 1 && f != REC_I1))
            return ((SVO_GP(const float))recs)[(size_t)(f * 16 + px) * a.rec_cap + i];
        else return (LDSF(dyn + lay.rec) + (f * 16 + px) * cap)[i];
    }
    __device__ inline float g_ld(int k, int i) const {       // sum g g^T: Gxx, Gxy, Gyy
        if constexpr (MODE != 0) return ((SVO_GP(const float))recs)[(size_t)(64 + k) * a.rec_cap + i];
        else return (LDSF(dyn + lay.kpf) + (KF_GXX + k) * cap)[i];
    }

    // ---- per-level records (sia_prep_kernel wrote them): HBM -> LDS, 16 B per lane and step
    __device__ void load_records(int slot) {
        const float* src = G(a.rec_ws) + (size_t)slot * SIA_REC_ROWS * a.rec_cap;
        recs = src;
        if constexpr (BIG) return;
        constexpr int n_rows = MODE == 1 ? 16 : 67;          // MODE 1: only the cost records (field REC_I1 = rows 0..15)
        // cap is a multiple of T, so a lane keeps its float4 column and walks the rows, 8 loads in flight
        const int c4 = cap >> 2;
        for (int col4 = threadIdx.x; col4 < c4; col4 += T) {
            for (int r0 = 0; r0 < n_rows; r0 += 8) {
                v4f v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (r0 + u < n_rows) v[u] = *reinterpret_cast<const v4f*>(src + (size_t)(r0 + u) * a.rec_cap + 4 * col4);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int row = r0 + u;
                    if (row < n_rows) {
                        SVO_LDS(float)* dst = row < 64 ? LDSF(dyn + lay.rec) + row * cap + 4 * col4
                                                       : LDSF(dyn + lay.kpf) + (KF_GXX + (row - 64)) * cap + 4 * col4;
                        *(SVO_LDS(v4f)*)dst = v[u];
                    }
                }
            }
        }
    }

    // ---- do_calc: project + get_total_intensity_diff
    __device__ float cost(const float pose[6]) {
        SIA_T(c0);
        PoseMats pm;
        pose_mats(pose, pm);
        SIA_T(c1);
        const CamD camd = make_camd(fx, fy, cx, cy, a.cam);
        const int ps = patch;
        const float half_size = ((float)ps - 1.0f) / 2.0f;
        float* buf = reinterpret_cast<float*>(dyn + lay.tbuf) + par * cap;
        par ^= 1;
        float total = 0;
        for (int i = threadIdx.x; i < npad; i += T) {     // (passes beyond the sequence's own keypoints would add exact zeros)
            float v = 0;
            if (kpf_ld(KF_ACT, i) != 0.f) {
                const svo_kp2d q = project_point(pm.Rd, pm.t, camd, svo_kp3d{kpf_ld(KF_PX, i), kpf_ld(KF_PY, i), kpf_ld(KF_PZ, i)});
                kpf_st(KF_QX, i, q.x); kpf_st(KF_QY, i, q.y);
                const float s2x = q.x - half_size, s2y = q.y - half_size;
                const float f2x = floorf(s2x), f2y = floorf(s2y);
                const float i1_0 = rec_ld(REC_I1, 0, i);
                // (absurd projections are kept out of the int conversion)
                if (i1_0 == i1_0 && f2x >= 0.f && f2y >= 0.f && f2x < 65536.f && f2y < 65536.f) {
                    const int ip2x = (int)f2x, ip2y = (int)f2y;
                    if (ip2y + ps < cur.h && ip2x + ps < cur.w) {
                        const float x22 = s2x - (float)ip2x, y22 = s2y - (float)ip2y;
                        const float x21 = 1.0f - x22, y21 = 1.0f - y22;
                        const float m0 = x21 * y21, m1 = x22 * y21, m2 = x21 * y22, m3 = x22 * y22;
                        const int o = ip2y * cur.stride + ip2x;
                        float b[5][5];
#pragma unroll
                        for (int r = 0; r < 5; r++)
#pragma unroll
                            for (int c = 0; c < 5; c++) b[r][c] = cur.at(o + r * cur.stride + c);
#pragma unroll
                        for (int px = 0; px < 16; px++) {
                            const int r = px >> 2, c = px & 3;
                            float i2 = 0;
                            i2 += m0 * b[r][c];
                            i2 += m1 * b[r][c + 1];
                            i2 += m2 * b[r + 1][c];
                            i2 += m3 * b[r + 1][c + 1];
                            v += fabsf(rec_ld(REC_I1, px, i) - i2);
                        }
                    }
                }
            }
            // diff += ... in keypoint order (inactive / outside: an exact 0): same bits in every lane
            if constexpr (WAVES == 1) total = ordered_wave_sum(v, total);
            else LDSF(buf)[i] = v;
        }
        SIA_T(c2);
        if constexpr (WAVES > 1) {
            __syncthreads();
            total = ordered_sum(buf, n, 0.f);
        }
        SIA_T(c3);
        SIA_ADD(1, c1, c0); SIA_ADD(2, c2, c1); SIA_ADD(3, c3, c2); SIA_ADD(4, 1, 0);
        return total;
    }

    // ---- get_gradient (with calculate_hessian) at the pose of the LAST cost evaluation
    // (the projections kept by cost() are reused); the step comes back in grad[6], in every lane
    __device__ void gradient(const float pose[6], float grad[6], float* dbg) {
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        SIA_T(g0);
        PoseMats pm;
        pose_mats(pose, pm);
        const bool exact = a.exact_pinv != 0;
        float acc[27];
#pragma unroll
        for (int q = 0; q < 27; q++) acc[q] = 0;
        // exact: accumulator of this lane (wave 0): H(r,c) += row_r * row_c, lanes 21..26: b_r -= row_r * diff
        int ia = 0, ib = 0;
        if (lane < 21) { ia = c_tri_r[lane]; ib = c_tri_c[lane]; }
        else if (lane < 27) { ia = lane - 21; ib = 6; }
        float eacc = 0;
        const float wlim = (float)(cur.w - 2), hlim = (float)(cur.h - 2);
        constexpr int KS = 20;                       // floats per keypoint in a staging plane (16 + pad: no write conflicts)
        constexpr int SIA_STG = sia_stg(T);
        constexpr int PS = SIA_STG * KS + 4;         // plane stride: the 7 planes start on different banks
        float* stage = reinterpret_cast<float*>(dyn + lay.stage);      // [7][PS], index slot*KS + px

        // J (2x6) and the 16 residuals of keypoint i (a lane's work of one pass)
        auto compute_kp = [&](int i, float (&J)[12], float (&d)[16], bool& active) {
#pragma unroll
            for (int q = 0; q < 12; q++) J[q] = 0;
            active = kpf_ld(KF_ACT, i) != 0.f;
            if (active) {
                float X[3] = {kpf_ld(KF_PX, i) - pm.t[0], kpf_ld(KF_PY, i) - pm.t[1], kpf_ld(KF_PZ, i) - pm.t[2]};
                mat33f_vec(pm.Ri, X, X);
                pose_jacobian(fx, fy, X[0], X[1], X[2], J);
            }
            // residuals: the same walk over the patch as above, from the projection. The 16 patch sums
            // read a 6x6 block of the current image: it is loaded up front (36 independent reads,
            // clamped to the image) and every patch sum whose taps start where the walk says they
            // should — all of them, unless a float increment rounds across an integer — takes its 3x3
            // taps from the block; the others read the image directly.
            {
                const float kx0 = kpf_ld(KF_QX, i) - 2.f, ky0 = kpf_ld(KF_QY, i) - 2.f;
                const int bx = (int)fminf(fmaxf(floorf(kx0 - 0.5f), -8.f), 65536.f);
                const int by = (int)fminf(fmaxf(floorf(ky0 - 0.5f), -8.f), 65536.f);
                float blk[6][6];
                if (active) {
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        const int ro = min(max(by + r, 0), cur.h - 1) * cur.stride;
#pragma unroll
                        for (int c = 0; c < 6; c++) blk[r][c] = cur.at(ro + min(max(bx + c, 0), cur.w - 1));
                    }
                }
                float kx = kx0, ky = ky0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int px = r * 4 + c;
                        float dd = 0;
                        if (active) {
                            const float psr = rec_ld(REC_PS, px, i);
                            // (kx - 1.0) < 0 || (ky - 1.0) < 0 || (kx + 2.0) > cols || (ky + 2.0) > rows of :449-454:
                            // the double sums are exact, so these float compares decide the same way
                            if (psr == psr &&                   // reference pixel inside (:449-451)
                                !(kx < 1.f || ky < 1.f || kx > wlim || ky > hlim)) {
                                // get_patch_sum (:82-112) at (kx, ky)
                                const float sx = kx - 0.5f, sy = ky - 0.5f;
                                const int ipx = (int)floorf(sx), ipy = (int)floorf(sy);
                                if (ipx == bx + c && ipy == by + r) {
                                    const float x2 = sx - (float)ipx, y2 = sy - (float)ipy;
                                    const float x1 = 1.0f - x2, y1 = 1.0f - y2;
                                    const float intensity =
                                        x1 * y1 * blk[r][c] + y1 * blk[r][c + 1] + x2 * y1 * blk[r][c + 2] +
                                        x1 * blk[r + 1][c] + blk[r + 1][c + 1] + x2 * blk[r + 1][c + 2] +
                                        x1 * y2 * blk[r + 2][c] + y2 * blk[r + 2][c + 1] + x2 * y2 * blk[r + 2][c + 2];
                                    dd = intensity - psr;
                                } else {
                                    dd = patch_sum_lds(cur, kx, ky) - psr;
                                }
                            }
                        }
                        d[px] = dd;
                        kx += 1.f;
                    }
                    kx -= 4.f;
                    ky += 1.f;
                }
            }
        };
        // one keypoint's rows of gradient_times_jacobians (:376-388) and its diffs into slot `slot` of a staging buffer
        auto stage_kp = [&](float* stg, int slot, int i, bool active, const float (&J)[12], const float (&d)[16]) {
#pragma unroll
            for (int p4 = 0; p4 < 4; p4++) {
                float g0[4], g1[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    g0[e] = active ? rec_ld(REC_G0, p4 * 4 + e, i) : 0.f;
                    g1[e] = active ? rec_ld(REC_G1, p4 * 4 + e, i) : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    v4f row;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float sum = 0;
                        sum += g0[e] * J[q];
                        sum += g1[e] * J[6 + q];
                        row[e] = sum;
                    }
                    *(SVO_LDS(v4f)*)(LDSF(stg) + q * PS + slot * KS + p4 * 4) = row;
                }
                // residual -= row * diff (:472-477) == residual += row * (-diff), exactly
                *(SVO_LDS(v4f)*)(LDSF(stg) + 6 * PS + slot * KS + p4 * 4) =
                    v4f{-d[p4 * 4], -d[p4 * 4 + 1], -d[p4 * 4 + 2], -d[p4 * 4 + 3]};
            }
        };
        // hessian += row^T row (lanes 0..20), residual += row * (-diff) (lanes 21..26) over the m keypoints of a
        // staging buffer: one multiply and one add of the chain per patch pixel, in storage order. SIA_ACC_U keypoints
        // per trip: their LDS reads are in flight together and the chain starts when the first arrive. (Slots past
        // m up to the next multiple of SIA_ACC_U were staged as zeros like every keypoint that takes no part: they
        // add exact zeros.) Products two at a time (v_pk_mul_f32: the same IEEE products), the adds in storage order.
        auto accumulate_chunk = [&](const float* stg, int m) {
            const SVO_LDS(float)* pa = LDSCF(stg) + ia * PS;
            const SVO_LDS(float)* pb = LDSCF(stg) + ib * PS;
            for (int j = 0; j < m; j += SIA_ACC_U) {
                v4f ra[SIA_ACC_U][4], rb[SIA_ACC_U][4];
#pragma unroll
                for (int u = 0; u < SIA_ACC_U; u++) {
                    const SVO_LDS(v4f)* qa = (const SVO_LDS(v4f)*)(pa + (j + u) * KS);
                    const SVO_LDS(v4f)* qb = (const SVO_LDS(v4f)*)(pb + (j + u) * KS);
#pragma unroll
                    for (int e = 0; e < 4; e++) { ra[u][e] = qa[e]; rb[u][e] = qb[e]; }
                }
#pragma unroll
                for (int u = 0; u < SIA_ACC_U; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const v2f p0 = ra[u][e].lo * rb[u][e].lo, p1 = ra[u][e].hi * rb[u][e].hi;
                        eacc += p0.x; eacc += p0.y; eacc += p1.x; eacc += p1.y;
                    }
            }
        };

        if (WAVES > 2 && exact) {
            // ---- reference-order accumulation as a pipeline (four waves per sequence): wave 0 only adds,
            // chunk after chunk of SIA_STG keypoints in index order; the other waves compute the keypoints
            // (Jacobian, residuals, rows) and fill the two staging buffers ahead of it. Hand-over through LDS
            // counters (a buffer's k-th fill may start when its (k-1)-th has been read). The chain of adds in
            // wave 0 — sequential by definition — is what a gradient call then costs.
            static_assert(WAVES <= 2 || SIA_STG == 64, "the pipelined accumulation hands over whole waves of keypoints");
            SVO_LDS(int)* cnt = (SVO_LDS(int)*)(dyn + lay.pipe);       // [0..1] fills done, [2..3] reads done, per buffer
            float* const stg0 = stage;
            float* const stg1 = stage + 7 * PS;
            if (tid < 4) cnt[tid] = 0;
            __syncthreads();
            const int n_chunks = (n + SIA_STG - 1) / SIA_STG;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: the two roles never share a wave)
            if (wave == 0) {
                for (int c = 0; c < n_chunks; c++) {
                    const int bsel = c & 1, k = c >> 1;
                    while (__hip_atomic_load(&cnt[bsel], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (lane < 27) accumulate_chunk(bsel ? stg1 : stg0, min(SIA_STG, n - c * SIA_STG));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_store(&cnt[2 + bsel], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                for (int c = wave - 1; c < n_chunks; c += WAVES - 1) {
                    const int i = c * SIA_STG + lane;
                    float J[12], d[16];
                    bool active;
#pragma unroll
                    for (int q = 0; q < 12; q++) J[q] = 0;
                    compute_kp(i, J, d, active);
                    const int bsel = c & 1, k = c >> 1;
                    while (__hip_atomic_load(&cnt[2 + bsel], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < k) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    stage_kp(bsel ? stg1 : stg0, lane, i, active, J, d);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_store(&cnt[bsel], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } else
        for (int i0 = 0; i0 < npad; i0 += T) {
            const int i = i0 + tid;
            float J[12], d[16];
            bool active;
#pragma unroll
            for (int q = 0; q < 12; q++) J[q] = 0;
            compute_kp(i, J, d, active);
            if (!exact) {
                float s0 = 0, s1 = 0, Gxx = 0, Gxy = 0, Gyy = 0;
                if (active) {                               // (slots past the keypoints hold no records)
#pragma unroll
                    for (int px = 0; px < 16; px++) {
                        s0 += rec_ld(REC_G0, px, i) * d[px];
                        s1 += rec_ld(REC_G1, px, i) * d[px];
                    }
                    Gxx = g_ld(0, i); Gxy = g_ld(1, i); Gyy = g_ld(2, i);
                }
                int q = 0;
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = r; c < 6; c++) {          // H(r,c) = J_r^T (G J)_c
                        const float mm0 = Gxx * J[c] + Gxy * J[6 + c], mm1 = Gxy * J[c] + Gyy * J[6 + c];
                        acc[q++] += J[r] * mm0 + J[6 + r] * mm1;
                    }
#pragma unroll
                for (int r = 0; r < 6; r++) acc[21 + r] -= J[r] * s0 + J[6 + r] * s1;
            } else {
                // SIA_STG keypoints at a time stage the rows of gradient_times_jacobians (:376-388) and the
                // diffs, keypoint-major; wave 0 adds them in storage order.
                for (int sub = 0; sub < T / SIA_STG; sub++) {
                    sia_sync<WAVES>();                      // the previous keypoints have been consumed
                    if ((tid / SIA_STG) == sub) stage_kp(stage, tid % SIA_STG, i, active, J, d);
                    sia_sync<WAVES>();
                    if (wave == 0 && lane < 27)
                        accumulate_chunk(stage, min(SIA_STG, n - (i0 + sub * SIA_STG)));   // keypoints of this chunk, in index order
                }
            }
        }

        SIA_T(g1);
        float* sums = reinterpret_cast<float*>(dyn + lay.sums);        // [WAVES][32]
        float H[36], b[6];
        if (!exact) {
#pragma unroll
            for (int q = 0; q < 27; q++) acc[q] = wave_sum_dpp(acc[q]);
            if constexpr (WAVES > 1) {
                __syncthreads();
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < 27; q++) LDSF(sums)[wave * 32 + q] = acc[q];
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 27; q++) {
                    float v = LDSCF(sums)[q];
#pragma unroll
                    for (int w = 1; w < WAVES; w++) v += LDSCF(sums)[w * 32 + q];
                    acc[q] = v;
                }
            }
        } else {
            sia_sync<WAVES>();
            if (wave == 0 && lane < 27) LDSF(sums)[lane] = eacc;
            sia_sync<WAVES>();
#pragma unroll
            for (int q = 0; q < 27; q++) acc[q] = LDSCF(sums)[q];
        }
        {
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = r; c < 6; c++) { H[r * 6 + c] = acc[q]; H[c * 6 + r] = acc[q]; q++; }
#pragma unroll
            for (int r = 0; r < 6; r++) b[r] = acc[21 + r];
        }
        SIA_T(g2);
        float delta[6], pg[6];
        if (WAVES > 2) {
            // four waves: wave 0 solves, the others wait at the barrier (the same 6x6 SVD in every wave would
            // only take VALU time from the kernels of the other sequence groups)
            if (wave == 0) {
                gn_solve6(H, b, delta, exact);
                exponential_map(delta, pg);
                mat33f_vec(pm.R, pg, grad);
                mat33f_vec(pm.R, pg + 3, grad + 3);
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < 6; q++) LDSF(sums)[WAVES * 32 + q] = grad[q];
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 6; q++) grad[q] = LDSCF(sums)[WAVES * 32 + q];
            __syncthreads();                             // (sums is written again by the next call)
        } else {
#ifdef SVO_SVD_ONE_LANE
        // experiment: the solve under an exec mask of one lane (less switching power), result broadcast
        if (lane == 0) gn_solve6(H, b, delta, exact);
#pragma unroll
        for (int q = 0; q < 6; q++) delta[q] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(delta[q])));
#else
        gn_solve6(H, b, delta, exact);
#endif
        exponential_map(delta, pg);
        mat33f_vec(pm.R, pg, grad);                  // pose_estimator.cpp:495-497
        mat33f_vec(pm.R, pg + 3, grad + 3);
        }
        SIA_T(g3);
        SIA_ADD(5, g1, g0); SIA_ADD(6, g2, g1); SIA_ADD(7, g3, g2); SIA_ADD(8, 1, 0);
#ifndef SVO_SIA_STAMPS
        if (dbg && tid == 0) {
            for (int q = 0; q < 36; q++) dbg[q] = H[q];
            for (int q = 0; q < 6; q++) { dbg[36 + q] = b[q]; dbg[42 + q] = grad[q]; }
        }
#else
        (void)dbg;
#endif
    }

    // stage one level image into LDS (rows padded to a dword)
    __device__ void stage_image(const ImgView im) {
        if constexpr (BIG) { cur = LevelImg<BIG>{im.data, im.w, im.h, im.stride}; return; }
        const int tid = threadIdx.x;
        uint8_t* sc = dyn + lay.img;
        const int ls = (im.w + 3) & ~3;
        if ((((uintptr_t)im.data | (uintptr_t)im.stride) & 3) == 0) {
            const int wd = im.w >> 2;                       // whole dwords per row
            // 64 lanes share a row segment (coalesced); a wave takes every WAVES-th group of 8 rows:
            // 8 loads in flight per lane, no divisions
            const int lane = tid & 63, wave = tid >> 6;
            for (int c0 = 0; c0 < wd; c0 += 64) {
                const int c = c0 + lane;
                for (int r0 = wave * 8; r0 < im.h; r0 += 8 * WAVES) {
                    uint32_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (c < wd && r0 + u < im.h)
                            v[u] = *reinterpret_cast<const uint32_t*>(im.g() + (size_t)(r0 + u) * im.stride + 4 * c);
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (c < wd && r0 + u < im.h) *(SVO_LDS(uint32_t)*)(sc + (r0 + u) * ls + 4 * c) = v[u];
                }
            }
            const int tail = im.w & 3;
            for (int i = tid; i < tail * im.h; i += T) {
                const int r = i / tail, c = wd * 4 + (i - r * tail);
                *(SVO_LDS(uint8_t)*)(sc + r * ls + c) = im.g()[(size_t)r * im.stride + c];
            }
        } else {
            for (int i = tid; i < im.w * im.h; i += T) {
                const int r = i / im.w, c = i - r * im.w;
                *(SVO_LDS(uint8_t)*)(sc + r * ls + c) = im.g()[(size_t)r * im.stride + c];
            }
        }
        cur = LevelImg<BIG>{sc, im.w, im.h, ls};
    }

    __device__ void run() {
        const int tid = threadIdx.x;
        patch = a.cam.window_size_pose_estimator;
        // active set and points (PoseEstimatorCallback ctor, :238-245)
        for (int i = tid; i < cap; i += T) {
            const bool active = i < n && !(a.flags && (G(a.flags)[i] & SVO_IGNORE_TEMPORARY));
            svo_kp3d P = {0, 0, 0};
            if (active) P = G(a.kps3d)[i];
            kpf_st(KF_PX, i, P.x); kpf_st(KF_PY, i, P.y); kpf_st(KF_PZ, i, P.z);
            kpf_st(KF_QX, i, 0.f); kpf_st(KF_QY, i, 0.f);
            kpf_st(KF_ACT, i, active ? 1.f : 0.f);
        }
        SIA_T(k0);
        float est[6];
#pragma unroll
        for (int j = 0; j < 6; j++) est[j] = G(a.pose_guess)[j];
        float last_cost = 0;
        bool dbg_done = false;

        for (int lv = a.cam.max_pyramid_levels; lv > a.cam.min_pyramid_level_pose_estimation; lv--) {
            const int level = lv - 1;
            const int divider = 1 << level;
            fx = a.cam.fx / divider; fy = a.cam.fy / divider;
            cx = a.cam.cx / divider; cy = a.cam.cy / divider;
            sia_sync<WAVES>();               // everybody is done with the previous level's image
            SIA_T(l0);
            stage_image(a.cur[level]);
            load_records(level - a.cam.min_pyramid_level_pose_estimation);
            sia_sync<WAVES>();
            SIA_T(l1);
            SIA_ADD(0, l1, l0);

            // ---- estimate_pose_at_level (:166-222); i is shared by both loops.
            // Every get_gradient(x0) directly follows the cost evaluation of x0 (the
            // initial one or the accepted trial), so it reuses that projection.
            const int maxIter = 50;
            float x0[6];
#pragma unroll
            for (int j = 0; j < 6; j++) x0[j] = est[j];
            int n_grad = 0, n_cost = 1, accepted = 0, exit_small = 0;
            float prev_cost = cost(x0);
            const float initial = prev_cost;
            for (int i = 0; i < maxIter; i++) {
                float* dbg = (a.dbg_H && !dbg_done && level == a.dbg_level) ? (float*)G(a.dbg_H) : nullptr;
                float g[6];
                gradient(x0, g, dbg);
                if (dbg) dbg_done = true;
                n_grad++;
                float k = 1.0f;
                for (; i < maxIter; i++) {
                    float x[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) x[j] = x0[j] + k * g[j];
                    const float new_cost = cost(x);
                    n_cost++;
                    if (new_cost < prev_cost) {
#pragma unroll
                        for (int j = 0; j < 6; j++) x0[j] = x[j];
                        prev_cost = new_cost;
                        accepted++;
                        break;
                    } else if ((double)fabsf(new_cost - prev_cost) < 1.0) {
                        i = maxIter;
                        exit_small = 1;
                        break;
                    } else
                        k /= 2;
                }
            }
#pragma unroll
            for (int j = 0; j < 6; j++) est[j] = x0[j];
            last_cost = prev_cost;
            if (tid == 0 && a.trace) {
                svo_gn_trace t;
                t.level = level; t.n_gradient = n_grad; t.n_cost = n_cost; t.n_accepted = accepted;
                t.exit_small = exit_small; t.initial_cost = initial; t.final_cost = prev_cost;
                for (int j = 0; j < 6; j++) t.pose[j] = x0[j];
                G(a.trace)[level] = t;
            }
        }
#ifdef SVO_SIA_STAMPS
        if (tid == 0 && a.dbg_H) {
            st[9] = __builtin_readcyclecounter() - k0;
            for (int j = 0; j < 12; j++) G(a.dbg_H)[j] = (float)st[j];
        }
#endif
        if (tid == 0) {
            for (int j = 0; j < 6; j++) G(a.pose_out)[j] = est[j];
            if (a.cost_out) *G(a.cost_out) = last_cost;
            if (a.mats_out) {    // once per sequence instead of once per keypoint workgroup of klt_track_kernel
                PoseMats pm;
                pose_mats(est, pm);
                *G(a.mats_out) = pm;
            }
        }
    }
};

// (diagnostic builds: -DSVO_SIA_OCC=n caps the registers at 512/n per lane, tools/build_variants.sh)
#ifdef SVO_SIA_OCC
#define SIA_OCC_ATTR __attribute__((amdgpu_waves_per_eu(SVO_SIA_OCC)))
#else
#define SIA_OCC_ATTR
#endif
template <int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES) SIA_OCC_ATTR void sia_gn_kernel(const SiaArgs* __restrict__ args, int img_bytes, int cap) {
    // One or two waves per sequence, a long serial chain: when the window kernels of another
    // sequence group share the SIMD, this wave should win the issue arbitration (it needs few slots)
    __builtin_amdgcn_s_setprio(3);
    const SiaArgs& a = args[blockIdx.x];
    const int n = min(*G(a.n_ptr), cap);
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    const SiaLds lay = sia_lds_layout(img_bytes, cap, 64 * WAVES, a.exact_pinv != 0, MODE);
    Sia<WAVES, MODE> s(a, n, cap, dyn, lay);
    s.run();
}

// largest level image the estimator uses (LDS rows padded to a dword)
static int sia_img_bytes(const svo_camera_settings& cam, int width, int height) {
    int best = 0;
    for (int lv = cam.max_pyramid_levels; lv > cam.min_pyramid_level_pose_estimation; lv--) {
        const int level = lv - 1;
        const int b = (((width >> level) + 3) & ~3) * (height >> level);
        if (b > best) best = b;
    }
    return best;
}

template <int WAVES, int MODE>
static bool sia_launch_shape(const SiaArgs* d_args, int batch, int img, int cap, size_t lds, hipStream_t stream) {
    static LdsLimit limit;
    if (raise_lds_limit(limit, reinterpret_cast<const void*>(sia_gn_kernel<WAVES, MODE>), (int)SIA_LDS_BUDGET) != hipSuccess)
        return false;
    hipLaunchKernelGGL((sia_gn_kernel<WAVES, MODE>), dim3(batch), dim3(64 * WAVES), lds, stream, d_args, img, cap);
    return true;
}

// Workgroup shape of a launch: `batch` sequences of at most n_bound keypoints, one keypoint per
// lane and pass: as many waves as 64-keypoint passes, at most 4. A few sequences keep everything
// in LDS (MODE 0: records 256 B per keypoint + the finest level image), a batch only the cost
// records (MODE 1); sets that do not fit (1920x1080, ~1700 keypoints) read records, per-keypoint
// values and image taps from L2 (MODE 2). Returns false if n_bound exceeds the workspaces.
bool launch_sia(const SiaArgs* d_args, int batch, const svo_camera_settings& cam, int width,
                int height, int n_bound, int rec_cap, int exact, hipStream_t stream) {
    const int nb = std::max(n_bound, 1);
    const int n_lv = cam.max_pyramid_levels - cam.min_pyramid_level_pose_estimation;
    const int img = sia_img_bytes(cam, width, height);
    // A batch of sequences: records, per-keypoint values and image taps from L2 (MODE 2) and one
    // wave per 128 keypoints, i.e. ~38 KB of LDS per sequence (the staging area of the ordered
    // accumulation) instead of 74-96 KB. With six and more sequence groups in flight LDS is what the
    // window kernels (SSD 17 KB, pyramid 15 KB, KLT 10 KB per workgroup) run short of: 1536 sequences
    // in 6 groups, frames/s: MODE 0 174 K, MODE 1 184 K, MODE 2 two waves 200 K, one wave 206 K.
    // (at most 4 waves: with 8 a wave may only hold 256 registers and the kernel spills ~80 into scratch)
    const bool batched = batch >= 32;
    int waves = nb <= 64 ? 1 : nb <= 128 ? 2 : 4;
    int mode = 0;
    // (the bound is the LARGEST keypoint set of the launch; a sequence only walks its own keypoints. One wave up to
    // 192: with keyframes at the reference's rate the sets reach ~160, and the two-wave shape costs every sequence
    // of the launch a second wave — 5.7 against 4.8 ms per launch in the round-3 profile)
    if (batched) { mode = 2; waves = nb <= 192 ? 1 : nb <= 384 ? 2 : 4; }
    // (experiments: SVO_SIA_MODE = 0 / 1 / 2 and SVO_SIA_WAVES = 1 / 2 / 4 force the shape of batched launches)
    static const int env_mode = getenv("SVO_SIA_MODE") ? atoi(getenv("SVO_SIA_MODE")) : -1;
    static const int env_waves = getenv("SVO_SIA_WAVES") ? atoi(getenv("SVO_SIA_WAVES")) : 0;
    if (batched && env_mode >= 0 && env_mode <= 2) mode = env_mode;
    if (batched && (env_waves == 1 || env_waves == 2 || env_waves == 4)) waves = env_waves;
    int T = 64 * waves;
    int cap = (nb + T - 1) / T * T;                   // every lane of every pass owns a slot
    size_t lds = sia_lds_layout(img, cap, T, exact != 0, mode).total;
    if (lds > SIA_LDS_BUDGET && mode == 0) {
        mode = 1;
        lds = sia_lds_layout(img, cap, T, exact != 0, mode).total;
    }
    if (lds > SIA_LDS_BUDGET && mode != 2) {
        mode = 2; waves = 4; T = 256;
        cap = (nb + T - 1) / T * T;
        lds = sia_lds_layout(img, cap, T, exact != 0, mode).total;
    }
    if (lds > SIA_LDS_BUDGET || cap > rec_cap) return false;
    hipLaunchKernelGGL(sia_prep_kernel, dim3((((nb + 3) & ~3) * 16 + 63) / 64, n_lv, batch), dim3(64), 0, stream, d_args);
#define SIA_CASE(W, M) if (waves == W && mode == M) return sia_launch_shape<W, M>(d_args, batch, img, cap, lds, stream);
    SIA_CASE(1, 0) SIA_CASE(2, 0) SIA_CASE(4, 0)
    SIA_CASE(1, 1) SIA_CASE(2, 1) SIA_CASE(4, 1)
    SIA_CASE(1, 2) SIA_CASE(2, 2) SIA_CASE(4, 2)
#undef SIA_CASE
    return false;
}

size_t sia_rec_ws_floats(const svo_camera_settings& cam, int rec_cap) {
    return (size_t)(cam.max_pyramid_levels - cam.min_pyramid_level_pose_estimation) * SIA_REC_ROWS * rec_cap;
}

}  // namespace svo
