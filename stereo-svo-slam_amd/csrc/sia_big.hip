// sia_big.hip — sparse image alignment for keypoint sets that do not fit the
// register-resident workgroup of sia.hip (more than 1024 keypoints per sequence,
// or level images larger than LDS: the 1920x1080 configuration). Same rows of
// SURVEY §8a and the same reference lines as sia.hip; per-pixel records live in an
// HBM workspace and 512 threads share a sequence, 16 lanes per keypoint patch.
// The float sums here are fixed-order trees, not the reference's sequential order.
#include "svo_kernels.hpp"
#include <atomic>
#include <algorithm>
#include "svo_reduce.hpp"

namespace svo {
namespace big {

#ifndef SVO_SIA_THREADS
#define SVO_SIA_THREADS 512
#endif
constexpr int SIA_THREADS = SVO_SIA_THREADS;
constexpr int SIA_WAVES = SIA_THREADS / 64;

struct SiaShared {
    PoseMats pm;
    float red[SIA_WAVES][28];
    float sums[28];
    float grad[6];
};


// position of patch pixel (r, c) exactly as the reference's nested loops reach
// it: x++ per column, x -= 4 and y++ at the end of a row (float arithmetic).
__device__ inline void patch_pos(float x0, float y0, int r, int c, float& x, float& y) {
    x = x0; y = y0;
    for (int rr = 0; rr < r; rr++) {
        x += 1.f; x += 1.f; x += 1.f; x += 1.f;
        x -= 4.f;
        y += 1.f;
    }
    for (int cc = 0; cc < c; cc++) x += 1.f;
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// one level image: the LDS copy (LDS == true, row stride = w) or the HBM original
template <bool LDS>
struct LevelImg {
    const uint8_t* p;
    int w, h, stride;
    __device__ inline int at(int y, int x) const { return (int)mem_ld<LDS>(p, y * stride + x); }
};

// get_patch_sum, src/lib/pose_estimator.cpp:82-112
template <bool LDS>
__device__ inline float patch_sum_img(const LevelImg<LDS>& im, float cx, float cy) {
    const float sx = cx - 0.5f, sy = cy - 0.5f;
    const int ipx = (int)floorf(sx), ipy = (int)floorf(sy);
    const float x2 = sx - (float)ipx, y2 = sy - (float)ipy;
    const float x1 = 1.0f - x2, y1 = 1.0f - y2;
    const int o = ipy * im.stride + ipx;
    const float a00 = (float)mem_ld<LDS>(im.p, o), a01 = (float)mem_ld<LDS>(im.p, o + 1),
                a02 = (float)mem_ld<LDS>(im.p, o + 2);
    const float a10 = (float)mem_ld<LDS>(im.p, o + im.stride), a11 = (float)mem_ld<LDS>(im.p, o + im.stride + 1),
                a12 = (float)mem_ld<LDS>(im.p, o + im.stride + 2);
    const float a20 = (float)mem_ld<LDS>(im.p, o + 2 * im.stride), a21 = (float)mem_ld<LDS>(im.p, o + 2 * im.stride + 1),
                a22 = (float)mem_ld<LDS>(im.p, o + 2 * im.stride + 2);
    const float intensity = x1 * y1 * a00 + y1 * a01 + x2 * y1 * a02 +
                            x1 * a10 + a11 + x2 * a12 +
                            x1 * y2 * a20 + y2 * a21 + x2 * y2 * a22;
    return intensity;
}

// per-level working set of the workgroup. LDS == true: images, records and the
// per-keypoint arrays are in LDS; false: they stay in HBM (large configurations).
template <bool LDS>
struct LevelCtx {
    LevelImg<LDS> cur;      // sampled by every iteration: LDS copy when the working set is in LDS
    LevelImg<false> prev;   // read once per level (reference patches): stays in HBM / L2
    float fx, fy, cx, cy;
    int patch;              // window_size_pose_estimator
    // per patch pixel (n*16): reference cost sample, gradients, reference patch sum
    float* rec_i1; float* rec_g0; float* rec_g1; float* rec_ps;
    // per keypoint: point + active flag, sum g g^T
    v4f* kp_pt; v4f* kp_G;
    v2f* proj;              // projection at the pose of the last evaluation (always LDS)
    v4f* kp_cw;             // bilinear weights of the cost patch at that projection
    int* kp_cb;             // byte offset of its first tap in the current level image, < 0: outside
    float* kp_J;            // [n][12] Jacobian of the last evaluated pose (gradient only)
    float* kp_rows;         // [n][8]  x and y of the four patch rows as the reference's loops reach them
};

__device__ inline float block_sum1(float v, SiaShared& sh) {
    v = wave_sum_dpp(v);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) sh.red[tid >> 6][0] = v;
    __syncthreads();
    float s = sh.red[0][0];
#pragma unroll
    for (int w = 1; w < SIA_WAVES; w++) s += sh.red[w][0];
    return s;   // same order in every thread
}

// do_calc: project + get_total_intensity_diff. Phase B (one thread per keypoint)
// projects and derives the bilinear weights / first tap of the 4x4 cost patch;
// phase C (one thread per patch pixel) is 1 record read, 4 taps and 8 flops.
template <bool LDS>
__device__ float sia_cost(const SiaArgs& a, int n, const LevelCtx<LDS>& L, const float pose[6], SiaShared& sh) {
    const int tid = threadIdx.x;
    __syncthreads();                      // previous readers of pm / proj / red are done
    if (tid == 0) pose_mats(pose, sh.pm);
    __syncthreads();
    const CamD camd = make_camd(L.fx, L.fy, L.cx, L.cy, a.cam);
    const int ps = L.patch;
    const float half_size = ((float)ps - 1.0f) / 2.0f;
    for (int i = tid; i < n; i += SIA_THREADS) {
        const v4f p = mem_ld<LDS>(L.kp_pt, i);
        int cb = -1;
        v4f cw = {0, 0, 0, 0};
        if (p.w != 0.f) {
            const svo_kp2d q = project_point(sh.pm.Rd, sh.pm.t, camd, svo_kp3d{p.x, p.y, p.z});
            mem_st<LDS>(L.proj, i, v2f{q.x, q.y});
            const float s2x = q.x - half_size, s2y = q.y - half_size;
            const float f2x = floorf(s2x), f2y = floorf(s2y);
            // (absurd projections are kept out of the int conversion)
            if (f2x >= 0.f && f2y >= 0.f && f2x < 65536.f && f2y < 65536.f) {
                const int ip2x = (int)f2x, ip2y = (int)f2y;
                if (ip2y + ps < L.cur.h && ip2x + ps < L.cur.w) {
                    const float x22 = s2x - (float)ip2x, y22 = s2y - (float)ip2y;
                    const float x21 = 1.0f - x22, y21 = 1.0f - y22;
                    cw = v4f{x21 * y21, x22 * y21, x21 * y22, x22 * y22};
                    cb = ip2y * L.cur.stride + ip2x;
                }
            }
        }
        mem_st<LDS>(L.kp_cw, i, cw);
        mem_st<LDS>(L.kp_cb, i, cb);
    }
    __syncthreads();
    float v = 0;
    for (int idx = tid; idx < n * 16; idx += SIA_THREADS) {
        const int kp = idx >> 4, px = idx & 15;
        const float i1 = mem_ld<LDS>(L.rec_i1, idx);
        const int cb = mem_ld<LDS>(L.kp_cb, kp);
        if (i1 != i1 || cb < 0) continue;            // reference or current half outside / inactive
        const v4f m = mem_ld<LDS>(L.kp_cw, kp);
        const int o = cb + (px >> 2) * L.cur.stride + (px & 3);
        float i2 = 0;
        i2 += m.x * (float)mem_ld<LDS>(L.cur.p, o);
        i2 += m.y * (float)mem_ld<LDS>(L.cur.p, o + 1);
        i2 += m.z * (float)mem_ld<LDS>(L.cur.p, o + L.cur.stride);
        i2 += m.w * (float)mem_ld<LDS>(L.cur.p, o + L.cur.stride + 1);
        v += fabsf(i1 - i2);
    }
    const float total = block_sum1(v, sh);
    return total;
}

// (r, c) of the 21 upper-triangle entries of H in the order they are stored
__constant__ int8_t c_tri_r[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
__constant__ int8_t c_tri_c[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};

// get_gradient (with calculate_hessian) at the pose of the LAST cost evaluation
// (rotation in sh.pm and projections in L.proj are reused); leaves the step in sh.grad.
// Phase B' (one thread per keypoint): Jacobian and the four row starts of the
// residual patch. Phase C' (one thread per patch pixel): residual * gradient,
// summed over the 16 lanes of the patch (one DPP row); then every lane owns two
// of the 27 outputs (21 entries of J^T G J, 6 of -J^T s).
template <bool LDS>
__device__ void sia_gradient(const SiaArgs& a, int n, const LevelCtx<LDS>& L, SiaShared& sh, float* dbg) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += SIA_THREADS) {
        const v4f pt = mem_ld<LDS>(L.kp_pt, i);
        float J[12];
        float rows[8];
        if (pt.w != 0.f) {
            float X[3] = {pt.x - sh.pm.t[0], pt.y - sh.pm.t[1], pt.z - sh.pm.t[2]};
            mat33f_vec(sh.pm.Ri, X, X);
            pose_jacobian(L.fx, L.fy, X[0], X[1], X[2], J);
            const v2f q = mem_ld<LDS>(L.proj, i);
            float x = q.x - 2.f, y = q.y - 2.f;      // x++ per column, x -= 4 and y++ per row
#pragma unroll
            for (int r = 0; r < 4; r++) {
                rows[r] = x; rows[4 + r] = y;
                x += 1.f; x += 1.f; x += 1.f; x += 1.f;
                x -= 4.f;
                y += 1.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 12; k++) J[k] = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) rows[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < 12; k++) mem_st<LDS>(L.kp_J, i * 12 + k, J[k]);
#pragma unroll
        for (int k = 0; k < 8; k++) mem_st<LDS>(L.kp_rows, i * 8 + k, rows[k]);
    }
    __syncthreads();
    float acc_a = 0, acc_b = 0;                      // outputs px and px + 16
    const int px = tid & 15;
    const int oa = px, ob = px + 16;
    const int ra = c_tri_r[oa], ca = c_tri_c[oa];
    const int rb = ob < 21 ? c_tri_r[ob] : ob - 21, cb = ob < 21 ? c_tri_c[ob] : 0;
    const int npad = (n * 16 + 63) & ~63;            // whole waves take part in the row sums
    for (int idx = tid; idx < npad; idx += SIA_THREADS) {
        const int kp = idx >> 4;
        float s0 = 0, s1 = 0;
        const bool in = idx < n * 16;
        if (in) {
            const float psr = mem_ld<LDS>(L.rec_ps, idx);
            if (psr == psr) {                        // active and reference pixel inside (:449-451)
                const int r = px >> 2, c = px & 3;
                float kx = mem_ld<LDS>(L.kp_rows, kp * 8 + r);
                const float ky = mem_ld<LDS>(L.kp_rows, kp * 8 + 4 + r);
                kx += (c > 0) ? 1.f : 0.f;           // x++ per column (adding 0 is exact)
                kx += (c > 1) ? 1.f : 0.f;
                kx += (c > 2) ? 1.f : 0.f;
                if (!(((double)kx - 1.0) < 0 || ((double)ky - 1.0) < 0 ||
                      ((double)kx + 2.0) > L.cur.w || ((double)ky + 2.0) > L.cur.h)) {
                    const float d = patch_sum_img(L.cur, kx, ky) - psr;
                    s0 = mem_ld<LDS>(L.rec_g0, idx) * d;
                    s1 = mem_ld<LDS>(L.rec_g1, idx) * d;
                }
            }
        }
        s0 = row16_sum_dpp(s0);
        s1 = row16_sum_dpp(s1);
        if (in) {
            const float* Jk = L.kp_J + kp * 12;
            const v4f G = mem_ld<LDS>(L.kp_G, kp);
            {   // H entry (ra, ca) = J_r^T (G J)_c
                const float j0c = mem_ld<LDS>(Jk, ca), j1c = mem_ld<LDS>(Jk, 6 + ca);
                const float m0 = G.x * j0c + G.y * j1c, m1 = G.y * j0c + G.z * j1c;
                acc_a += mem_ld<LDS>(Jk, ra) * m0 + mem_ld<LDS>(Jk, 6 + ra) * m1;
            }
            if (ob < 21) {
                const float j0c = mem_ld<LDS>(Jk, cb), j1c = mem_ld<LDS>(Jk, 6 + cb);
                const float m0 = G.x * j0c + G.y * j1c, m1 = G.y * j0c + G.z * j1c;
                acc_b += mem_ld<LDS>(Jk, rb) * m0 + mem_ld<LDS>(Jk, 6 + rb) * m1;
            } else if (ob < 27) {
                acc_b -= mem_ld<LDS>(Jk, rb) * s0 + mem_ld<LDS>(Jk, 6 + rb) * s1;
            }
        }
    }
    // the four patches of a wave, then the waves
    acc_a += __shfl_xor(acc_a, 16, 64); acc_a += __shfl_xor(acc_a, 32, 64);
    acc_b += __shfl_xor(acc_b, 16, 64); acc_b += __shfl_xor(acc_b, 32, 64);
    if (lane < 16) {
        sh.red[wave][oa] = acc_a;
        if (ob < 27) sh.red[wave][ob] = acc_b;
    }
    __syncthreads();
    if (tid < 27) {
        float s = sh.red[0][tid];
#pragma unroll
        for (int w = 1; w < SIA_WAVES; w++) s += sh.red[w][tid];
        sh.sums[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float H[36], b[6], delta[6], pg[6];
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = r; c < 6; c++) { H[r * 6 + c] = sh.sums[q]; H[c * 6 + r] = sh.sums[q]; q++; }
#pragma unroll
        for (int r = 0; r < 6; r++) b[r] = sh.sums[21 + r];
        gn_solve6(H, b, delta, a.exact_pinv != 0);
        exponential_map(delta, pg);
        mat33f_vec(sh.pm.R, pg, sh.grad);            // pose_estimator.cpp:495-497
        mat33f_vec(sh.pm.R, pg + 3, sh.grad + 3);
        if (dbg) {
            for (int k = 0; k < 36; k++) dbg[k] = H[k];
            for (int k = 0; k < 6; k++) { dbg[36 + k] = b[k]; dbg[42 + k] = sh.grad[k]; }
        }
    }
    __syncthreads();
}

// Working set of one sequence, sized by the number of keypoints n:
//   per keypoint 128 B : projection, cost weights + offset, Jacobian, row starts,
//                        point + active flag, sum g g^T
//   per patch pixel 16 B (256 B per keypoint): the per-level records
//   2 x the largest level image the estimator uses
// It lives in LDS when it fits the budget (n <= ~250 at 752x480), else in the
// HBM workspace (SiaArgs::kp_ws / cache). Host and device share this function.
struct SiaLds {
    size_t proj, kp_cw, kp_cb, kp_J, kp_rows, kp_pt, kp_G, img_cur, rec, total;
    int img_bytes;
};

__host__ __device__ inline SiaLds sia_lds_layout(int n, int max_img_bytes) {
    SiaLds l;
    const size_t np = ((size_t)n + 15) & ~(size_t)15;
    size_t off = 0;
    l.proj = off;    off += np * 8;
    l.kp_cb = off;   off += np * 4;
    l.kp_cw = off;   off += np * 16;
    l.kp_J = off;    off += np * 48;
    l.kp_rows = off; off += np * 32;
    l.kp_pt = off;   off += np * 16;
    l.kp_G = off;    off += np * 16;
    const size_t img = ((size_t)max_img_bytes + 15) & ~(size_t)15;
    l.img_bytes = (int)img;
    l.img_cur = off; off += img;
    l.rec = off;     off += np * 256;
    l.total = off;
    return l;
}

template <bool LDS>
__device__ void sia_run(const SiaArgs& a, int n, SiaShared& sh, uint8_t* dyn, const SiaLds& lay) {
    const int tid = threadIdx.x;
    LevelCtx<LDS> L;
    {
        // LDS: the dynamic segment; HBM: kp_ws (cap * 40 floats) and cache (cap * 16 float4)
        uint8_t* base = LDS ? dyn : reinterpret_cast<uint8_t*>(a.kp_ws);
        const SiaLds g = LDS ? lay : sia_lds_layout(a.cap, 0);
        L.proj = reinterpret_cast<v2f*>(base + g.proj);
        L.kp_cw = reinterpret_cast<v4f*>(base + g.kp_cw);
        L.kp_cb = reinterpret_cast<int*>(base + g.kp_cb);
        L.kp_J = reinterpret_cast<float*>(base + g.kp_J);
        L.kp_rows = reinterpret_cast<float*>(base + g.kp_rows);
        L.kp_pt = reinterpret_cast<v4f*>(base + g.kp_pt);
        L.kp_G = reinterpret_cast<v4f*>(base + g.kp_G);
        float* r = LDS ? reinterpret_cast<float*>(dyn + lay.rec) : reinterpret_cast<float*>(a.cache);
        const size_t c16 = LDS ? (size_t)(((size_t)n + 15) & ~(size_t)15) * 16 : (size_t)a.cap * 16;
        L.rec_i1 = r; L.rec_g0 = r + c16; L.rec_g1 = r + 2 * c16; L.rec_ps = r + 3 * c16;
    }
    L.patch = a.cam.window_size_pose_estimator;

    // active set and points (PoseEstimatorCallback ctor, :238-245)
    for (int i = tid; i < n; i += SIA_THREADS) {
        const svo_kp3d P = a.kps3d[i];
        const bool active = !(a.flags && (a.flags[i] & SVO_IGNORE_TEMPORARY));
        mem_st<LDS>(L.kp_pt, i, v4f{P.x, P.y, P.z, active ? 1.f : 0.f});
    }

    float est[6];
#pragma unroll
    for (int j = 0; j < 6; j++) est[j] = a.pose_guess[j];
    float last_cost = 0;
    bool dbg_done = false;

    for (int lv = a.cam.max_pyramid_levels; lv > a.cam.min_pyramid_level_pose_estimation; lv--) {
        const int level = lv - 1;
        const int divider = 1 << level;
        const ImgView cur = a.cur[level], prev = a.prev[level];
        L.fx = a.cam.fx / divider; L.fy = a.cam.fy / divider;
        L.cx = a.cam.cx / divider; L.cy = a.cam.cy / divider;
        __syncthreads();                 // everybody is done with the previous level's LDS
        if (LDS) {
            uint8_t* sc = dyn + lay.img_cur;
            for (int i = tid; i < cur.w * cur.h; i += SIA_THREADS) {
                const int r = i / cur.w, c = i - r * cur.w;
                mem_st<true>(sc, i, cur.data[(size_t)r * cur.stride + c]);
            }
            L.cur = LevelImg<LDS>{sc, cur.w, cur.h, cur.w};
        } else {
            L.cur = LevelImg<LDS>{cur.data, cur.w, cur.h, cur.stride};
        }
        L.prev = LevelImg<false>{prev.data, prev.w, prev.h, prev.stride};
        __syncthreads();

        // ---- per-level records that depend on the previous frame only
        const int npad = (n * 16 + 63) & ~63;
        for (int idx = tid; idx < npad; idx += SIA_THREADS) {
            const int kp = idx >> 4, px = idx & 15;
            float g0 = 0, g1 = 0, psr = __builtin_nanf(""), i1 = __builtin_nanf("");
            bool active = false;
            svo_kp2d kref = {0, 0};
            if (idx < n * 16) {
                active = mem_ld<LDS>(L.kp_pt, kp).w != 0.f;
                kref = a.kps2d[kp];
                if (level != 0) { kref.x /= divider; kref.y /= divider; }      // setLevel
            }
            if (active) {
                float kx, ky;
                patch_pos(kref.x - 2.f, kref.y - 2.f, px >> 2, px & 3, kx, ky);
                // calculate_hessian bounds (:351-352)
                if (!(((double)kx - 2.0) < 0 || ((double)ky - 2.0) < 0 ||
                      ((double)kx + 3.0) >= L.prev.w || ((double)ky + 3.0) >= L.prev.h)) {
                    const float int1 = patch_sum_img(L.prev, kx + 1, ky);
                    const float int2 = patch_sum_img(L.prev, kx - 1, ky);
                    const float int3 = patch_sum_img(L.prev, kx, ky + 1);
                    const float int4 = patch_sum_img(L.prev, kx, ky - 1);
                    g0 = int1 - int2; g1 = int3 - int4;
                }
                // reference half of the residual test (:449-453)
                if (!(((double)kx - 1.0) < 0 || ((double)ky - 1.0) < 0 ||
                      ((double)kx + 2.0) > L.prev.w || ((double)ky + 2.0) > L.prev.h))
                    psr = patch_sum_img(L.prev, kx, ky);
                // reference half of the cost (image_comparison.cpp:20-88)
                const int ps = L.patch;
                const float half_size = ((float)ps - 1.0f) / 2.0f;
                const float s1x = kref.x - half_size, s1y = kref.y - half_size;
                const float f1x = floorf(s1x), f1y = floorf(s1y);
                if (f1x >= 0.f && f1y >= 0.f && f1x < 65536.f && f1y < 65536.f) {
                    const int ip1x = (int)f1x, ip1y = (int)f1y;
                    if (ip1y + ps < L.prev.h && ip1x + ps < L.prev.w) {
                        const float x12 = s1x - (float)ip1x, y12 = s1y - (float)ip1y;
                        const float x11 = 1.0f - x12, y11 = 1.0f - y12;
                        const float m0 = x11 * y11, m1 = x12 * y11, m2 = x11 * y12, m3 = x12 * y12;
                        const int yy = (px >> 2) + ip1y, xx = (px & 3) + ip1x;
                        float t = 0;
                        t += m0 * (float)L.prev.at(yy, xx);
                        t += m1 * (float)L.prev.at(yy, xx + 1);
                        t += m2 * (float)L.prev.at(yy + 1, xx);
                        t += m3 * (float)L.prev.at(yy + 1, xx + 1);
                        i1 = t;
                    }
                }
            }
            if (idx < n * 16) {
                mem_st<LDS>(L.rec_i1, idx, i1); mem_st<LDS>(L.rec_g0, idx, g0);
                mem_st<LDS>(L.rec_g1, idx, g1); mem_st<LDS>(L.rec_ps, idx, psr);
            }
            const float gxx = row16_sum_dpp(g0 * g0), gxy = row16_sum_dpp(g0 * g1),
                        gyy = row16_sum_dpp(g1 * g1);
            if (px == 0 && idx < n * 16) mem_st<LDS>(L.kp_G, kp, v4f{gxx, gxy, gyy, 0.f});
        }
        __syncthreads();

        // ---- estimate_pose_at_level (:166-222); i is shared by both loops.
        // Every get_gradient(x0) directly follows the cost evaluation of x0 (the
        // initial one or the accepted trial), so it reuses that rotation/projection.
        const int maxIter = 50;
        float x0[6];
#pragma unroll
        for (int j = 0; j < 6; j++) x0[j] = est[j];
        int n_grad = 0, n_cost = 1, accepted = 0, exit_small = 0;
        float prev_cost = sia_cost<LDS>(a, n, L, x0, sh);
        const float initial = prev_cost;
        for (int i = 0; i < maxIter; i++) {
            float* dbg = (a.dbg_H && !dbg_done && level == a.dbg_level) ? a.dbg_H : nullptr;
            sia_gradient<LDS>(a, n, L, sh, dbg);
            if (dbg) dbg_done = true;
            n_grad++;
            float g[6];
#pragma unroll
            for (int j = 0; j < 6; j++) g[j] = sh.grad[j];
            float k = 1.0f;
            for (; i < maxIter; i++) {
                float x[6];
#pragma unroll
                for (int j = 0; j < 6; j++) x[j] = x0[j] + k * g[j];
                const float new_cost = sia_cost<LDS>(a, n, L, x, sh);
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
            a.trace[level] = t;
        }
    }
    if (tid == 0) {
        for (int j = 0; j < 6; j++) a.pose_out[j] = est[j];
        if (a.cost_out) *a.cost_out = last_cost;
        if (a.mats_out) {    // once per sequence instead of once per keypoint workgroup of klt_track_kernel
            PoseMats pm;
            pose_mats(est, pm);
            *a.mats_out = pm;
        }
    }
}

__global__ __launch_bounds__(SIA_THREADS) void sia_gn_big_kernel(const SiaArgs* __restrict__ args,
                                                              int max_img_bytes, int lds_bytes) {
    const SiaArgs& a = args[blockIdx.x];
    const int n = min(*a.n_ptr, a.cap);
    __shared__ SiaShared sh;
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    const SiaLds lay = sia_lds_layout(n, max_img_bytes);
    // the whole working set in LDS when this frame's keypoints fit, else the HBM workspace
    if (max_img_bytes > 0 && lay.total <= (size_t)lds_bytes) sia_run<true>(a, n, sh, dyn, lay);
    else sia_run<false>(a, n, sh, dyn, lay);
}

}  // namespace big

void launch_sia_big(const SiaArgs* d_args, int batch, const svo_camera_settings& cam, int width,
                    int height, int n_bound, hipStream_t stream) {
    (void)cam; (void)width; (void)height; (void)n_bound;
    // working set in the HBM workspace (SiaArgs::kp_ws / cache): max_img_bytes = 0
    hipLaunchKernelGGL(big::sia_gn_big_kernel, dim3(batch), dim3(big::SIA_THREADS), 0, stream, d_args, 0, 0);
}

}  // namespace svo
