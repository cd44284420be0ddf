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
//   * per level the current-frame level image is staged into LDS (it is
//     23x15 .. 188x120 bytes at 752x480), so the 4x4 / 3x3 taps of every
//     evaluation are LDS reads;
//   * everything that depends only on the previous frame — image gradients,
//     reference patch sums, the reference half of the cost — is computed once
//     per level into a 16-byte record per patch pixel (coalesced float4);
//   * an evaluation = Rodrigues (lane 0) -> projection in double, one thread
//     per keypoint -> 16 lanes per keypoint patch -> wave/LDS reduction of the
//     cost or of the 21+6 normal-equation sums -> 6x6 pseudo-inverse,
//     exponential map and the accept / halve / stop decision on lane 0.
// Since the reference never caches its Hessian (the member is shadowed,
// pose_estimator.cpp:399 vs :61) J^T J is rebuilt per get_gradient call; here
// it is factored as sum_kp J_kp^T (sum_px g g^T) J_kp with the inner 2x2
// constant per level.
#include "svo_kernels.hpp"
#include "svo_reduce.hpp"

namespace svo {

constexpr int SIA_THREADS = 1024;
constexpr size_t SIA_LDS_BUDGET = 140 * 1024;

struct SiaShared {
    PoseMats pm;
    float red[SIA_THREADS / 64][32];
    float sums[32];
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

struct LevelCtx {
    const uint8_t* cur;     // LDS copy or global
    int cur_stride, cur_w, cur_h;
    ImgView prev;
    float fx, fy, cx, cy;
    int patch;              // window_size_pose_estimator
};

// do_calc: project + get_total_intensity_diff
__device__ float sia_cost(const SiaArgs& a, int n, const LevelCtx& L, const float pose[6],
                          SiaShared& sh, svo_kp2d* s_proj) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) pose_mats(pose, sh.pm);
    __syncthreads();
    const CamD camd = make_camd(L.fx, L.fy, L.cx, L.cy, a.cam);
    for (int i = tid; i < n; i += SIA_THREADS)
        if (a.kp_ws[(size_t)i * 8 + 5] != 0.f)
            s_proj[i] = project_point(sh.pm.Rd, sh.pm.t, camd, a.kps3d[i]);
    __syncthreads();
    float v[1] = {0};
    const int ps = L.patch;
    for (int idx = tid; idx < n * 16; idx += SIA_THREADS) {
        const int kp = idx >> 4, px = idx & 15;
        const float i1 = a.cache[idx].w;
        if (i1 != i1) continue;                      // reference half invalid / inactive
        const svo_kp2d q = s_proj[kp];
        const float half_size = ((float)ps - 1.0f) / 2.0f;
        const float s2x = q.x - half_size, s2y = q.y - half_size;
        const float f2x = floorf(s2x), f2y = floorf(s2y);
        // keep absurd projections out of the int conversion
        if (!(f2x >= 0.f && f2y >= 0.f && f2x < 65536.f && f2y < 65536.f)) continue;
        const int ip2x = (int)f2x, ip2y = (int)f2y;
        if (!(ip2y + ps < L.cur_h && ip2x + ps < L.cur_w)) continue;
        const float x22 = s2x - (float)ip2x, y22 = s2y - (float)ip2y;
        const float x21 = 1.0f - x22, y21 = 1.0f - y22;
        const float m0 = x21 * y21, m1 = x22 * y21, m2 = x21 * y22, m3 = x22 * y22;
        const int i = px >> 2, j = px & 3;
        const uint8_t* p = L.cur + (long)(i + ip2y) * L.cur_stride + ip2x + j;
        float i2 = 0;
        i2 += m0 * (float)p[0];
        i2 += m1 * (float)p[1];
        i2 += m2 * (float)p[L.cur_stride];
        i2 += m3 * (float)p[L.cur_stride + 1];
        v[0] += fabsf(i1 - i2);
    }
    block_reduce<1, SIA_THREADS>(v, sh.red, sh.sums);
    return sh.sums[0];
}

// get_gradient (with calculate_hessian): leaves the 6-vector step in sh.grad
__device__ void sia_gradient(const SiaArgs& a, int n, const LevelCtx& L, const float pose[6],
                             SiaShared& sh, svo_kp2d* s_proj, float* dbg) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) pose_mats(pose, sh.pm);
    __syncthreads();
    const CamD camd = make_camd(L.fx, L.fy, L.cx, L.cy, a.cam);
    for (int i = tid; i < n; i += SIA_THREADS)
        if (a.kp_ws[(size_t)i * 8 + 5] != 0.f)
            s_proj[i] = project_point(sh.pm.Rd, sh.pm.t, camd, a.kps3d[i]);
    __syncthreads();
    float v[27];
#pragma unroll
    for (int k = 0; k < 27; k++) v[k] = 0;
    const int npad = (n * 16 + 63) & ~63;            // whole waves take part in the shuffles
    for (int idx = tid; idx < npad; idx += SIA_THREADS) {
        const int kp = idx >> 4, px = idx & 15;
        float s0 = 0, s1 = 0;
        bool active = false;
        if (idx < n * 16) {
            active = a.kp_ws[(size_t)kp * 8 + 5] != 0.f;
            const float4 rec = a.cache[idx];
            if (active && rec.z == rec.z) {          // reference pixel inside (:449-451)
                const svo_kp2d q = s_proj[kp];
                float kx, ky;
                patch_pos(q.x - 2.f, q.y - 2.f, px >> 2, px & 3, kx, ky);
                if (!(((double)kx - 1.0) < 0 || ((double)ky - 1.0) < 0 ||
                      ((double)kx + 2.0) > L.cur_w || ((double)ky + 2.0) > L.cur_h)) {
                    const float d = patch_sum(L.cur, L.cur_stride, kx, ky) - rec.z;
                    s0 = rec.x * d;
                    s1 = rec.y * d;
                }
            }
        }
        // sum over the 16 pixels of the patch (16 consecutive lanes)
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64);
            s1 += __shfl_xor(s1, o, 64);
        }
        if (px == 0 && active) {
            const svo_kp3d P = a.kps3d[kp];
            float X[3] = {P.x - sh.pm.t[0], P.y - sh.pm.t[1], P.z - sh.pm.t[2]};
            mat33f_vec(sh.pm.Ri, X, X);
            float J[12];
            pose_jacobian(L.fx, L.fy, X[0], X[1], X[2], J);
            const float gxx = a.kp_ws[(size_t)kp * 8 + 0], gxy = a.kp_ws[(size_t)kp * 8 + 1],
                        gyy = a.kp_ws[(size_t)kp * 8 + 2];
            float M0[6], M1[6];                      // (sum g g^T) J
#pragma unroll
            for (int k = 0; k < 6; k++) {
                M0[k] = gxx * J[k] + gxy * J[6 + k];
                M1[k] = gxy * J[k] + gyy * J[6 + k];
            }
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = r; c < 6; c++) v[q++] += J[r] * M0[c] + J[6 + r] * M1[c];
#pragma unroll
            for (int k = 0; k < 6; k++) v[21 + k] -= J[k] * s0 + J[6 + k] * s1;
        }
    }
    block_reduce<27, SIA_THREADS>(v, sh.red, sh.sums);
    if (tid == 0) {
        float H[36], b[6], delta[6], pg[6];
        int q = 0;
        for (int r = 0; r < 6; r++)
            for (int c = r; c < 6; c++) { H[r * 6 + c] = sh.sums[q]; H[c * 6 + r] = sh.sums[q]; q++; }
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

__global__ __launch_bounds__(SIA_THREADS) void sia_gn_kernel(const SiaArgs* __restrict__ args,
                                                              unsigned lds_img_bytes) {
    const SiaArgs& a = args[blockIdx.x];
    const int n = min(*a.n_ptr, a.cap);
    const int tid = threadIdx.x;
    __shared__ SiaShared sh;
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    svo_kp2d* s_proj = reinterpret_cast<svo_kp2d*>(dyn);
    uint8_t* s_img = dyn + (((size_t)a.cap * sizeof(svo_kp2d) + 15) & ~(size_t)15);

    float est[6];
    for (int j = 0; j < 6; j++) est[j] = a.pose_guess[j];
    float last_cost = 0;
    bool dbg_done = false;

    for (int lv = a.cam.max_pyramid_levels; lv > a.cam.min_pyramid_level_pose_estimation; lv--) {
        const int level = lv - 1;
        const int divider = 1 << level;
        LevelCtx L;
        L.prev = a.prev[level];
        const ImgView cur = a.cur[level];
        L.fx = a.cam.fx / divider; L.fy = a.cam.fy / divider;
        L.cx = a.cam.cx / divider; L.cy = a.cam.cy / divider;
        L.patch = a.cam.window_size_pose_estimator;
        L.cur_w = cur.w; L.cur_h = cur.h;
        __syncthreads();
        if ((size_t)cur.w * cur.h <= lds_img_bytes) {
            for (int i = tid; i < cur.w * cur.h; i += SIA_THREADS) {
                const int r = i / cur.w, c = i - r * cur.w;
                s_img[i] = cur.data[(size_t)r * cur.stride + c];
            }
            L.cur = s_img; L.cur_stride = cur.w;
        } else {
            L.cur = cur.data; L.cur_stride = cur.stride;
        }

        // ---- per-level records that depend on the previous frame only
        const int npad = (n * 16 + 63) & ~63;
        for (int idx = tid; idx < npad; idx += SIA_THREADS) {
            const int kp = idx >> 4, px = idx & 15;
            float g0 = 0, g1 = 0, psr = __builtin_nanf(""), i1 = __builtin_nanf("");
            bool active = false;
            svo_kp2d kref = {0, 0};
            if (idx < n * 16) {
                active = !(a.flags && (a.flags[kp] & SVO_IGNORE_TEMPORARY));   // ctor, :238-245
                kref = a.kps2d[kp];
                if (level != 0) { kref.x /= divider; kref.y /= divider; }      // setLevel
            }
            if (active) {
                float kx, ky;
                patch_pos(kref.x - 2.f, kref.y - 2.f, px >> 2, px & 3, kx, ky);
                // calculate_hessian bounds (:351-352)
                if (!(((double)kx - 2.0) < 0 || ((double)ky - 2.0) < 0 ||
                      ((double)kx + 3.0) >= L.prev.w || ((double)ky + 3.0) >= L.prev.h)) {
                    const float int1 = patch_sum(L.prev.data, L.prev.stride, kx + 1, ky);
                    const float int2 = patch_sum(L.prev.data, L.prev.stride, kx - 1, ky);
                    const float int3 = patch_sum(L.prev.data, L.prev.stride, kx, ky + 1);
                    const float int4 = patch_sum(L.prev.data, L.prev.stride, kx, ky - 1);
                    g0 = int1 - int2; g1 = int3 - int4;
                }
                // reference half of the residual test (:449-453)
                if (!(((double)kx - 1.0) < 0 || ((double)ky - 1.0) < 0 ||
                      ((double)kx + 2.0) > L.prev.w || ((double)ky + 2.0) > L.prev.h))
                    psr = patch_sum(L.prev.data, L.prev.stride, kx, ky);
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
                        const int i = px >> 2, j = px & 3;
                        const uint8_t* p = L.prev.data + (size_t)(i + ip1y) * L.prev.stride + ip1x + j;
                        float t = 0;
                        t += m0 * (float)p[0];
                        t += m1 * (float)p[1];
                        t += m2 * (float)p[L.prev.stride];
                        t += m3 * (float)p[L.prev.stride + 1];
                        i1 = t;
                    }
                }
            }
            if (idx < n * 16) a.cache[idx] = make_float4(g0, g1, psr, i1);
            float gxx = g0 * g0, gxy = g0 * g1, gyy = g1 * g1;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {
                gxx += __shfl_xor(gxx, o, 64);
                gxy += __shfl_xor(gxy, o, 64);
                gyy += __shfl_xor(gyy, o, 64);
            }
            if (px == 0 && idx < n * 16) {
                float* w = a.kp_ws + (size_t)kp * 8;
                w[0] = gxx; w[1] = gxy; w[2] = gyy; w[3] = kref.x; w[4] = kref.y;
                w[5] = active ? 1.f : 0.f;
            }
        }
        __syncthreads();

        // ---- estimate_pose_at_level (:166-222); i is shared by both loops
        const int maxIter = 50;
        float x0[6];
        for (int j = 0; j < 6; j++) x0[j] = est[j];
        int n_grad = 0, n_cost = 1, accepted = 0, exit_small = 0;
        float prev_cost = sia_cost(a, n, L, x0, sh, s_proj);
        const float initial = prev_cost;
        for (int i = 0; i < maxIter; i++) {
            float* dbg = (a.dbg_H && !dbg_done && level == a.dbg_level) ? a.dbg_H : nullptr;
            sia_gradient(a, n, L, x0, sh, s_proj, dbg);
            if (dbg) dbg_done = true;
            n_grad++;
            float g[6];
            for (int j = 0; j < 6; j++) g[j] = sh.grad[j];
            float k = 1.0f;
            for (; i < maxIter; i++) {
                float x[6];
                for (int j = 0; j < 6; j++) x[j] = x0[j] + k * g[j];
                const float new_cost = sia_cost(a, n, L, x, sh, s_proj);
                n_cost++;
                if (new_cost < prev_cost) {
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
    }
}

size_t sia_lds_bytes(const svo_camera_settings& cam, int width, int height, int cap) {
    const size_t proj = ((size_t)cap * sizeof(svo_kp2d) + 15) & ~(size_t)15;
    size_t img = 0;
    for (int lv = cam.max_pyramid_levels; lv > cam.min_pyramid_level_pose_estimation; lv--) {
        const int level = lv - 1;
        const size_t b = (size_t)(width >> level) * (size_t)(height >> level);
        if (proj + b <= SIA_LDS_BUDGET && b > img) img = b;
    }
    return proj + img;
}

void launch_sia(const SiaArgs* d_args, int batch, size_t lds_bytes, int cap, hipStream_t stream) {
    static size_t configured = 0;
    if (lds_bytes > configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sia_gn_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        configured = lds_bytes;
    }
    const size_t proj = ((size_t)cap * sizeof(svo_kp2d) + 15) & ~(size_t)15;
    const unsigned img_bytes = (unsigned)(lds_bytes > proj ? lds_bytes - proj : 0);
    hipLaunchKernelGGL(sia_gn_kernel, dim3(batch), dim3(SIA_THREADS), lds_bytes, stream, d_args,
                       img_bytes);
}

}  // namespace svo
