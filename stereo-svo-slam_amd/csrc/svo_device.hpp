// svo_device.hpp — device-side math shared by the gfx950 kernels.
//
// The float arithmetic follows the reference expression by expression
// (built with -ffp-contract=off: no fused multiply-add), so per-element
// results match a CPU evaluation of the same formulas; only the order of the
// big reductions differs. Reference lines are cited per function
// (paths relative to the reference repository).
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/svo_types.h"
#include "../../include/svo_libm.h"

namespace svo {

// Pointers that reach a kernel through an argument block in memory are generic to the compiler:
// it emits flat_load / flat_store and must drain vmcnt AND lgkmcnt around every LDS access next to
// them. Device code therefore casts them to the global address space where they are used
// (G(ptr), ImgView::g()): global_load / global_store with counted waits, scalar loads for
// wave-uniform values.
#if defined(__HIP_DEVICE_COMPILE__)
#define SVO_GP(T) __attribute__((address_space(1))) T*
#else
#define SVO_GP(T) T*      /* host pass: device functions are only parsed */
#endif
template <class T>
__device__ __forceinline__ SVO_GP(T) G(T* p) { return (SVO_GP(T))p; }

struct ImgView {
    const uint8_t* data;
    int w, h, stride;
    __device__ __forceinline__ SVO_GP(const uint8_t) g() const { return (SVO_GP(const uint8_t))data; }
    __device__ __forceinline__ SVO_GP(uint8_t) gw() const { return (SVO_GP(uint8_t))data; }
};

__host__ __device__ inline ImgView make_view(const svo_image& im) {
    return ImgView{im.data, im.width, im.height, im.stride};
}

// ------------------------------------------------------ address-space typed access
// A generic pointer that is known to point into LDS: cast to address space 3 so
// the compiler emits ds_read/ds_write instead of flat instructions.
#define SVO_LDS(T) __attribute__((address_space(3))) T
template <bool LDS, typename T>
__device__ inline T mem_ld(const T* p, int i) {
    if constexpr (LDS) return ((const SVO_LDS(T)*)p)[i];
    else return p[i];
}
template <bool LDS, typename T>
__device__ inline void mem_st(T* p, int i, const T& v) {
    if constexpr (LDS) ((SVO_LDS(T)*)p)[i] = v;
    else p[i] = v;
}

// ---------------------------------------------------------------- wave ops
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline long long wave_sum_ll(long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// DPP (data-parallel primitive) cross-lane adds: no LDS crossbar, ~1 VALU op each.
// A DPP row is 16 lanes; controls: quad_perm(1,0,3,2)=0xB1, quad_perm(2,3,0,1)=0x4E,
// row_half_mirror=0x141, row_mirror=0x140.
template <int CTRL>
__device__ inline float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ inline int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
// sum over each group of 16 consecutive lanes, result in all 16 lanes (fixed order)
__device__ inline float row16_sum_dpp(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}
__device__ inline int row16_sum_dpp_i(int v) {
    v += dpp_i<0xB1>(v);
    v += dpp_i<0x4E>(v);
    v += dpp_i<0x141>(v);
    v += dpp_i<0x140>(v);
    return v;
}
// sum over the 64 lanes of the wave, same value (and same rounding) in every lane
__device__ inline float wave_sum_dpp(float v) {
    v = row16_sum_dpp(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return ((r0 + r1) + r2) + r3;
}
__device__ inline int wave_sum_dpp_i(int v) {
    v = row16_sum_dpp_i(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
// sum over the 64 lanes with the wave-wide DPP forms of gfx9 (row_bcast:15 / row_bcast:31 carry a
// row's last lane into the next rows): six fused v_add_u32_dpp and one v_readlane (wave_sum_dpp_i:
// four adds, four readlanes, three scalar adds). Integer adds: any order gives the same bits.
__device__ inline int wave_sum_bcast_i(int v) {
    v = row16_sum_dpp_i(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
// exact sum of per-lane int32 partials (|v| < 2^31) as a double (|sum| < 2^37 is an integer a double
// holds exactly): low 16 bits and the signed high part are reduced separately in int32. (float) of
// the result rounds the exact integer once, like (float) of the 64-bit integer sum.
__device__ inline double wave_sum_i32_to_f64(int v) {
    const int lo = wave_sum_bcast_i(v & 0xFFFF);
    const int hi = wave_sum_bcast_i(v >> 16);
    return (double)hi * 65536.0 + (double)lo;
}
// exact 64-bit sum of per-lane int32 partials (|v| < 2^31): low 16 bits and the
// signed high part are reduced separately in int32 and recombined
__device__ inline long long wave_sum_i32_to_i64(int v) {
    const int lo = wave_sum_dpp_i(v & 0xFFFF);
    const int hi = wave_sum_dpp_i(v >> 16);
    return (long long)hi * 65536LL + (long long)lo;
}

// ------------------------------------------------- reference-order sums
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// sequential sum of buf[0..n) continuing from s. buf is in LDS, zero padded to a multiple of 128
// floats (adding the zeros is exact): every block of 128 is 32 ds_read_b128 issued back to back
// and one chain of 128 adds behind them, so the add chain, not the LDS, sets the pace.
__device__ inline float ordered_sum(const float* buf, int n, float s) {
    const SVO_LDS(v4f)* p = (const SVO_LDS(v4f)*)buf;
    for (int b = 0; b < n; b += 128, p += 32) {
        v4f v[32];
#pragma unroll
        for (int j = 0; j < 32; j++) v[j] = p[j];
#pragma unroll
        for (int j = 0; j < 32; j++) { s += v[j].x; s += v[j].y; s += v[j].z; s += v[j].w; }
    }
    return s;
}

// sequential sum over the 64 lanes of a wave, lane 0 first (v_readlane + add: no LDS), continuing from s
__device__ inline float ordered_wave_sum(float v, float s) {
#pragma unroll
    for (int j = 0; j < 64; j++) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
    return s;
}

// ------------------------------------------------------------- rotations
// cv::Rodrigues (vector -> matrix) in double: PoseManager::set_pose,
// src/lib/pose_manager.cpp:15-16, and inside cv::projectPoints,
// src/lib/transform_keypoints.cpp:45. R(-r) is exactly the transpose.
__device__ inline void rodrigues_d(const float r[3], double R[9]) {
    double rx = r[0], ry = r[1], rz = r[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = 0.0;
        R[0] = R[4] = R[8] = 1.0;
        return;
    }
    double s, c;
    svo_sincos(theta, &s, &c);   // the same polynomial as the oracle: include/svo_libm.h
    const double c1 = 1.0 - c;
    const double itheta = 1.0 / theta;
    rx *= itheta; ry *= itheta; rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const double e = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
        double t = c * e;
        t = t + c1 * rrt[k];
        R[k] = t + s * r_x[k];
    }
}

// Everything a kernel needs about one pose: R(r) in double (R(-r) = transpose),
// the float copies PoseManager caches, and the translation.
struct PoseMats {
    double Rd[9];   // R(r)
    float R[9];     // (float)R(r)      get_rotation_matrix()
    float Ri[9];    // (float)R(-r)     get_inv_rotation_matrix()
    float t[3];
};

__device__ inline void pose_mats(const float pose[6], PoseMats& m) {
    const float r[3] = {pose[3], pose[4], pose[5]};
    rodrigues_d(r, m.Rd);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            m.R[i * 3 + j] = (float)m.Rd[i * 3 + j];
            m.Ri[i * 3 + j] = (float)m.Rd[j * 3 + i];
        }
    m.t[0] = pose[0]; m.t[1] = pose[1]; m.t[2] = pose[2];
}

// Matx33f * Vec3f: s = 0; s += a[k]*v[k]
__device__ inline void mat33f_vec(const float* a, const float v[3], float out[3]) {
    float t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) s += a[i * 3 + k] * v[k];
        t[i] = s;
    }
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}

// project_keypoints, src/lib/transform_keypoints.cpp:11-48: translate by -t in
// float, then cv::projectPoints(rvec = -r, tvec = 0) in double, float store.
struct CamD {
    double fx, fy, cx, cy, k1, k2, p1, p2, k3;
};
__device__ inline CamD make_camd(float fx, float fy, float cx, float cy, const svo_camera_settings& c) {
    return CamD{(double)fx, (double)fy, (double)cx, (double)cy, (double)c.k1, (double)c.k2,
                (double)c.p1, (double)c.p2, (double)c.k3};
}
__device__ inline svo_kp2d project_point(const double Rd[9], const float t[3], const CamD& c,
                                         const svo_kp3d P) {
    const float Xf = P.x - t[0], Yf = P.y - t[1], Zf = P.z - t[2];
    const double X = Xf, Y = Yf, Z = Zf;
    // rows of R(-r) are columns of R(r)
    double x = Rd[0] * X + Rd[3] * Y + Rd[6] * Z + 0.0;
    double y = Rd[1] * X + Rd[4] * Y + Rd[7] * Z + 0.0;
    double z = Rd[2] * X + Rd[5] * Y + Rd[8] * Z + 0.0;
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + c.k1 * r2 + c.k2 * r4 + c.k3 * r6;
    const double xd = x * cdist + c.p1 * a1 + c.p2 * a2;
    const double yd = y * cdist + c.p1 * a3 + c.p2 * a1;
    svo_kp2d o;
    o.x = (float)(xd * c.fx + c.cx);
    o.y = (float)(yd * c.fy + c.cy);
    return o;
}

// 2x6 Jacobian of src/lib/pose_estimator.cpp:343-344 / pose_refinement.cpp:380-381
__device__ inline void pose_jacobian(float fx, float fy, float x, float y, float z, float J[12]) {
    J[0] = -fx / z;  J[1] = 0;       J[2] = fx * x / (z * z);
    J[3] = fx * x * y / (z * z);     J[4] = -fx * (1 + (x * x) / (z * z)); J[5] = fx * y / z;
    J[6] = 0;        J[7] = -fy / z; J[8] = fy * y / (z * z);
    J[9] = fy * (1 + (y * y) / (z * z)); J[10] = -fy * x * y / (z * z);    J[11] = -fy * x / z;
}

// exponential_map, src/include/exponential_map.hpp:12-37 (norm fixed to 1,
// double scale factors applied to a float matrix).
__device__ inline void exponential_map(const float twist[6], float out[6]) {
    const float v[3] = {twist[0], twist[1], twist[2]};
    const float w[3] = {twist[3], twist[4], twist[5]};
    const float K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    float K2[9], M[9];
    const float norm = 1.0f;
    // 1 - cos(1.0) and 1.0 - sin(1.0) in double (the device math library would not fold them)
    const double c1 = 0x1.d6bafe095f2e8p-2;
    const double c2 = 0x1.44aadc3dbcc48p-3;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += K[i * 3 + k] * K[k * 3 + j];
            K2[i * 3 + j] = s;
        }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float e = (k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f;
        const float t0 = e * norm;
        const float t1 = (float)(K[k] * c1);
        const float t2 = (float)(K2[k] * c2);
        M[k] = (t0 + t1) + t2;
    }
    float t[3];
    mat33f_vec(M, v, t);
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
    out[3] = w[0]; out[4] = w[1]; out[5] = w[2];
}

// ------------------------------------------------------------- small SVD
// One-sided Jacobi SVD on the rows of At (OpenCV JacobiSVDImpl_<float>):
// float data, double dot products. Reached in the reference through
// Matx66f::inv(DECOMP_SVD) (pose_estimator.cpp:405, pose_refinement.cpp:398),
// cv::solve(DECOMP_SVD) (depth_filter.cpp:200) and cv::KalmanFilter::correct.
template <int M, int N>
__host__ __device__ inline void jacobi_svd(float (&At)[N][M], float (&W)[N], float (&Vt)[N][N]) {
    const float eps = FLT_EPSILON * 2;
    double Wd[N];
    for (int i = 0; i < N; i++) {
        double sd = 0;
        for (int k = 0; k < M; k++) { const float t = At[i][k]; sd += (double)t * t; }
        Wd[i] = sd;
        for (int k = 0; k < N; k++) Vt[i][k] = 0;
        Vt[i][i] = 1;
    }
    const int max_iter = M > 30 ? M : 30;
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < N - 1; i++)
            for (int j = i + 1; j < N; j++) {
                double a = Wd[i], p = 0, b = Wd[j];
                for (int k = 0; k < M; k++) p += (double)At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = svo_hypot(p, beta);
                float c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (float)sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (int k = 0; k < M; k++) {
                    const float t0 = c * At[i][k] + s * At[j][k];
                    const float t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                Wd[i] = a; Wd[j] = b;
                changed = true;
                for (int k = 0; k < N; k++) {
                    const float t0 = c * Vt[i][k] + s * Vt[j][k];
                    const float t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0; Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < N; i++) {
        double sd = 0;
        for (int k = 0; k < M; k++) { const float t = At[i][k]; sd += (double)t * t; }
        Wd[i] = sqrt(sd);
    }
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        for (int k = i + 1; k < N; k++)
            if (Wd[j] < Wd[k]) j = k;
        if (i != j) {
            const double tw = Wd[i]; Wd[i] = Wd[j]; Wd[j] = tw;
            for (int k = 0; k < M; k++) { const float t = At[i][k]; At[i][k] = At[j][k]; At[j][k] = t; }
            for (int k = 0; k < N; k++) { const float t = Vt[i][k]; Vt[i][k] = Vt[j][k]; Vt[j][k] = t; }
        }
    }
    for (int i = 0; i < N; i++) {
        W[i] = (float)Wd[i];
        const double sd = Wd[i];
        const float s = (float)(sd > (double)FLT_MIN ? 1 / sd : 0.);
        for (int k = 0; k < M; k++) At[i][k] *= s;
    }
}

// The same one-sided Jacobi SVD for the 6x6 systems of the Gauss-Newton kernels, written so
// that every array index is a compile-time constant (all loops over rows / columns / pairs are
// unrolled, the selection sort swaps under predicates): At, Vt and W live in registers instead of
// scratch memory. Same operations in the same order as jacobi_svd<6,6>: same bits.
// (The product of two floats is exact in double, so fma(x, y, acc) == acc + x * y bit for bit: the
// sums of squares and the dot products below use one instruction per term.)
__device__ inline void jacobi_svd6_reg(float (&At)[6][6], float (&W)[6], float (&Vt)[6][6]) {
    const float eps = FLT_EPSILON * 2;
    double Wd[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) { const float t = At[i][k]; sd = __builtin_fma((double)t, (double)t, sd); }
        Wd[i] = sd;
#pragma unroll
        for (int k = 0; k < 6; k++) Vt[i][k] = (i == k) ? 1.f : 0.f;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = i + 1; j < 6; j++) {
                double a = Wd[i], p = 0, b = Wd[j];
#pragma unroll
                for (int k = 0; k < 6; k++) p = __builtin_fma((double)At[i][k], (double)At[j][k], p);
                // |p| <= eps * sqrt(a*b) of the reference, decided without the square root whenever
                // p^2 and eps^2 a b are further apart than any rounding error of either side (1e-9
                // relative, the errors are ~1e-16); the exact expression decides the rest (and NaNs)
                const double pp = p * p, ee = ((double)eps * (double)eps) * (a * b);
                bool rotate = pp > ee * (1 + 1e-9);
                if (!rotate && !(pp < ee * (1 - 1e-9))) rotate = !(fabs(p) <= eps * sqrt(a * b));
                if (rotate) {
                    p *= 2;
                    const double beta = a - b, gamma = svo_hypot(p, beta);
                    // beta < 0: s = sqrt((gamma - beta) / 2 / gamma), c = p / (2 gamma s); else c and s swap
                    // roles with (gamma + beta) / (2 gamma): one square root and one division serve both
                    const bool neg = beta < 0;
                    const double num = neg ? (gamma - beta) * 0.5 : (gamma + beta);
                    const double den = neg ? gamma : gamma * 2;
                    const float r1 = (float)sqrt(num / den);
                    const float r2 = (float)(p / (gamma * r1 * 2));
                    const float c = neg ? r2 : r1, s = neg ? r1 : r2;
                    a = b = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const float t0 = c * At[i][k] + s * At[j][k];
                        const float t1 = -s * At[i][k] + c * At[j][k];
                        At[i][k] = t0; At[j][k] = t1;
                        a = __builtin_fma((double)t0, (double)t0, a); b = __builtin_fma((double)t1, (double)t1, b);
                    }
                    Wd[i] = a; Wd[j] = b;
                    changed = true;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const float t0 = c * Vt[i][k] + s * Vt[j][k];
                        const float t1 = -s * Vt[i][k] + c * Vt[j][k];
                        Vt[i][k] = t0; Vt[j][k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) { const float t = At[i][k]; sd = __builtin_fma((double)t, (double)t, sd); }
        Wd[i] = sqrt(sd);
    }
    // selection sort, largest first: row i <-> the first maximum of rows i..5
#pragma unroll
    for (int i = 0; i < 5; i++) {
        int j = i;
        double wj = Wd[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++)
            if (wj < Wd[k]) { j = k; wj = Wd[k]; }
#pragma unroll
        for (int k = i + 1; k < 6; k++)
            if (j == k) {
                const double tw = Wd[i]; Wd[i] = Wd[k]; Wd[k] = tw;
#pragma unroll
                for (int q = 0; q < 6; q++) { const float t = At[i][q]; At[i][q] = At[k][q]; At[k][q] = t; }
#pragma unroll
                for (int q = 0; q < 6; q++) { const float t = Vt[i][q]; Vt[i][q] = Vt[k][q]; Vt[k][q] = t; }
            }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        W[i] = (float)Wd[i];
        const double sd = Wd[i];
        const float s = (float)(sd > (double)FLT_MIN ? 1 / sd : 0.);
#pragma unroll
        for (int k = 0; k < 6; k++) At[i][k] *= s;
    }
}

// Matx66f::inv(DECOMP_SVD): zeros when sigma_max < FLT_EPSILON or
// sigma_min / sigma_max == 0, else V diag(1/w) U^T with the SVBkSb threshold.
__device__ inline void inv_svd6(const float H[36], float Hinv[36]) {
    float At[6][6], Vt[6][6], W[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) At[i][j] = H[j * 6 + i];
    jacobi_svd6_reg(At, W, Vt);
#pragma unroll
    for (int i = 0; i < 36; i++) Hinv[i] = 0;
    double threshold = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) threshold += W[i];
    threshold *= (float)(DBL_EPSILON * 2);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double wi = W[i];
        if (!(fabs(wi) <= threshold)) {
            wi = 1 / wi;
            double buffer[6];
#pragma unroll
            for (int j = 0; j < 6; j++) buffer[j] = At[i][j] * wi;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const float sv = Vt[i][r];
#pragma unroll
                for (int j = 0; j < 6; j++) Hinv[r * 6 + j] = (float)(Hinv[r * 6 + j] + sv * buffer[j]);
            }
        }
    }
    const bool ok = W[0] >= FLT_EPSILON ? (W[5] / W[0] != 0) : false;
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 36; i++) Hinv[i] = 0;
    }
}

// delta = pinv(H) b for the Gauss-Newton steps (pose_estimator.cpp:405,484;
// pose_refinement.cpp:398-399). Fast path: when H is comfortably positive
// definite its pseudo-inverse is its inverse, so H delta = b is solved by an
// LDL^T factorisation in double (~150 flops on one lane instead of a Jacobi
// SVD). Anything else (a pivot below 1e-6 of the largest diagonal entry, i.e.
// rank deficient at float precision, or H = 0) takes the SVD route of the
// reference, which is also what `exact` forces for every call.
__device__ inline void gn_solve6(const float H[36], const float b[6], float delta[6], bool exact) {
    bool ok = !exact;
    if (ok) {
        double L[6][6], D[6], dmax = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) dmax = fmax(dmax, (double)H[i * 6 + i]);
        const double tiny = dmax * 1e-6;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double d = H[j * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
            if (!(d > tiny) || !(dmax > 0)) ok = false;
            D[j] = d;
            const double id = 1.0 / d;
#pragma unroll
            for (int i = j + 1; i < 6; i++) {
                double v = H[i * 6 + j];
#pragma unroll
                for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
                L[i][j] = v * id;
            }
        }
        if (ok) {
            double z[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double v = b[i];
#pragma unroll
                for (int k = 0; k < i; k++) v -= L[i][k] * z[k];
                z[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 6; i++) z[i] /= D[i];
#pragma unroll
            for (int i = 5; i >= 0; i--) {
                double v = z[i];
#pragma unroll
                for (int k = i + 1; k < 6; k++) v -= L[k][i] * z[k];
                z[i] = v;
                delta[i] = (float)v;
            }
            return;
        }
    }
    float Hinv[36];
    inv_svd6(H, Hinv);
#pragma unroll
    for (int r = 0; r < 6; r++) {
        float sacc = 0;
#pragma unroll
        for (int c = 0; c < 6; c++) sacc += Hinv[r * 6 + c] * b[c];
        delta[r] = sacc;
    }
}

// cv::solve(A[3x2], b, x, DECOMP_SVD), src/lib/depth_filter.cpp:194-200
__device__ inline void solve_svd_3x2(const float A[6], const float b[3], float x[2]) {
    float At[2][3], Vt[2][2], W[2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) At[i][j] = A[j * 2 + i];
    jacobi_svd<3, 2>(At, W, Vt);
    x[0] = x[1] = 0;
    double threshold = ((double)W[0] + (double)W[1]) * (float)(DBL_EPSILON * 2);
    for (int i = 0; i < 2; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < 3; j++) s += At[i][j] * b[j];
        s *= wi;
        for (int j = 0; j < 2; j++) x[j] = (float)(x[j] + s * Vt[i][j]);
    }
}

// 1-state cv::KalmanFilter predict()+correct(meas) with A = H = 1
// (src/lib/depth_filter.cpp:202-215); gain through the 1x1 SVD solve.
__device__ inline void kf1_update(float& x, float& P, float Q, float R, float meas) {
    const float statePre = (float)((double)1.0f * (double)x);
    const float temp1 = (float)((double)1.0f * (double)P);
    const float errorCovPre = (float)((double)temp1 * (double)1.0f + (double)Q);
    const float temp2 = (float)((double)1.0f * (double)errorCovPre);
    const float temp3 = (float)((double)temp2 * (double)1.0f + (double)R);
    // 1x1 Jacobi SVD: w = |a|, u = a * (float)(1/w), v = 1
    float gain = 0;
    {
        const double sd = sqrt((double)temp3 * temp3);
        const float w = (float)sd;
        const float u = temp3 * (float)(sd > (double)FLT_MIN ? 1 / sd : 0.);
        const double threshold = (double)w * (float)(DBL_EPSILON * 2);
        if (!(fabs((double)w) <= threshold)) {
            const double wi = 1 / (double)w;
            double s = 0;
            s += u * temp2;
            s *= wi;
            gain = (float)(0.0f + s * 1.0f);
        }
    }
    const float temp5 = meas - (float)((double)1.0f * (double)statePre);
    x = (float)((double)gain * (double)temp5 + (double)statePre);
    P = (float)(-((double)gain * (double)temp2) + (double)errorCovPre);
}

// get_patch_sum, src/lib/pose_estimator.cpp:82-112 (3x3 taps = area sum of a
// 2x2 box centred at `c`; callers guard the bounds)
__device__ inline float patch_sum(const uint8_t* img, int stride, float cx, float cy) {
    const float sx = cx - 0.5f, sy = cy - 0.5f;
    const float fx_ = floorf(sx), fy_ = floorf(sy);
    const int ipx = (int)fx_, ipy = (int)fy_;
    const float x2 = sx - (float)ipx, y2 = sy - (float)ipy;
    const float x1 = 1.0f - x2, y1 = 1.0f - y2;
    const uint8_t* s1 = img + (long)ipy * stride + ipx;
    const uint8_t* s2 = s1 + stride;
    const uint8_t* s3 = s2 + stride;
    const float intensity = x1 * y1 * (float)s1[0] + y1 * (float)s1[1] + x2 * y1 * (float)s1[2] +
                            x1 * (float)s2[0] + (float)s2[1] + x2 * (float)s2[2] +
                            x1 * y2 * (float)s3[0] + y2 * (float)s3[1] + x2 * y2 * (float)s3[2];
    return intensity;
}

__device__ inline int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

}  // namespace svo
