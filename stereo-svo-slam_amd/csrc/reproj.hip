// reproj.hip — rows B1 and B3 of SURVEY §8a: the merge step of
// PoseRefiner::refine_pose (src/lib/pose_refinement.cpp:125-150) and the
// reprojection Gauss-Newton PoseRefiner::update_pose (:236-290) with
// PoseRefinerCallback::do_calc (:321-348) / get_gradient (:350-412).
//
// One persistent workgroup per sequence runs the whole line-search GN on the
// device — no host round trip per iteration. The group is ONE wavefront when
// many sequences share the launch (no barrier anywhere, 256 sequences fit 16
// CUs) and four wavefronts for a lone sequence.
//
// Every float reduction runs in the reference's order: the cost is the
// sequential sum over the keypoints (:328-341), the 36 + 6 normal-equation
// sums are sequential `hessian += J^T J`, `err += J^T diff` (:393-395). A
// lane computes the terms of its keypoints, stages them in LDS in keypoint
// order, and the accumulators (one lane each) walk that array in order. The
// stop test of the line search (|dcost| < 1e-4, :273) sits at float rounding
// of the cost, so any other order changes the accept / halve sequence.
// Rotation matrices, the 6x6 solve and the accept logic are wave-uniform and
// computed redundantly by every lane (no broadcast, no barrier).
#include "svo_kernels.hpp"
#include <algorithm>
#include <mutex>

namespace svo {


// CH keypoints are staged per step: one per thread (4 KB of LDS for the one-wave shape, 16 KB for
// four waves — LDS is what the window kernels of other sequence groups are short of)
template <int CH>
struct alignas(16) ReprojShared {
    float tbuf[2][CH];            // cost terms of a chunk, keypoint order (double buffered)
    float js[14][CH];             // J (2x6), diff (2) of a chunk
    float sums[28];
};

typedef v4f rp_v4f;

// The keypoints of the sequence, staged once in dynamic LDS (struct of arrays, `cap` entries each):
// point x y z, observed position x y, and 1 / 0 for "takes part" (none of the ignore flags).
struct RpKps {
    SVO_LDS(float)* f;
    int cap;
    __device__ inline SVO_LDS(float)* at(int field) const { return f + field * cap; }
};

template <int WAVES>
__device__ inline void rp_sync() {
    if constexpr (WAVES > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
}

// cost of PoseRefinerCallback::do_calc
template <int WAVES>
__device__ float reproj_cost(const ReprojArgs& a, const RpKps& kp, int n, const float pose[6], ReprojShared<64 * WAVES>& sh, int& par) {
    constexpr int T = 64 * WAVES, RP_CHUNK = T;
    const int tid = threadIdx.x;
    PoseMats pm;
    pose_mats(pose, pm);
    const CamD camd = make_camd(a.cam.fx, a.cam.fy, a.cam.cx, a.cam.cy, a.cam);
    float tot = 0;
    for (int c0 = 0; c0 < n; c0 += RP_CHUNK) {
        float* buf = sh.tbuf[par];
        par ^= 1;
        for (int j = tid; j < RP_CHUNK; j += T) {
            const int i = c0 + j;
            float t = 0;
            if (i < n && kp.at(5)[i] != 0.f) {
                const svo_kp2d q = project_point(pm.Rd, pm.t, camd, svo_kp3d{kp.at(0)[i], kp.at(1)[i], kp.at(2)[i]});
                const float d0 = fabsf(q.x - kp.at(3)[i]), d1 = fabsf(q.y - kp.at(4)[i]);
                t = d0 + d1;
            }
            // tot_diff += ... in keypoint order, skipped keypoints add an exact 0 (every lane, same bits)
            if constexpr (WAVES == 1) tot = ordered_wave_sum(t, tot);
            else ((SVO_LDS(float)*)buf)[j] = t;
        }
        if constexpr (WAVES > 1) {
            __syncthreads();
            tot = ordered_sum(buf, min(RP_CHUNK, n - c0), tot);
        }
    }
    return tot;
}

// get_gradient at `pose`: leaves the step in grad[6] (every lane)
template <int WAVES>
__device__ void reproj_gradient(const ReprojArgs& a, const RpKps& kp, int n, const float pose[6], ReprojShared<64 * WAVES>& sh, float grad[6]) {
    constexpr int T = 64 * WAVES, RP_CHUNK = T;
    const int tid = threadIdx.x;
    PoseMats pm;
    pose_mats(pose, pm);
    const CamD camd = make_camd(a.cam.fx, a.cam.fy, a.cam.cx, a.cam.cy, a.cam);
    const float fx = a.cam.fx, fy = a.cam.fy;
    // accumulator of this lane: s = js[i0]*js[i2] + js[i1]*js[i3]
    //   lanes 0..20  H(r,c), r <= c : J[r]*J[c] + J[6+r]*J[6+c]   (:393)
    //   lanes 21..26 err(r)        : J[r]*d0   + J[6+r]*d1       (:395)
    int i0 = 0, i1 = 6, i2 = 0, i3 = 6;
    if (tid < 21) {
        int r = 0, c = tid;
        while (c >= 6 - r) { c -= 6 - r; r++; }
        c += r;
        i0 = r; i1 = 6 + r; i2 = c; i3 = 6 + c;
    } else if (tid < 27) {
        const int r = tid - 21;
        i0 = r; i1 = 6 + r; i2 = 12; i3 = 13;
    }
    float acc = 0;
    for (int c0 = 0; c0 < n; c0 += RP_CHUNK) {
        rp_sync<WAVES>();                          // the previous chunk has been consumed
        for (int j = tid; j < RP_CHUNK; j += T) {
            const int i = c0 + j;
            float J[12], d0 = 0, d1 = 0;
#pragma unroll
            for (int k = 0; k < 12; k++) J[k] = 0;
            if (i < n && kp.at(5)[i] != 0.f) {
                const svo_kp3d P = svo_kp3d{kp.at(0)[i], kp.at(1)[i], kp.at(2)[i]};
                const svo_kp2d q = project_point(pm.Rd, pm.t, camd, P);
                float X[3] = {P.x - pm.t[0], P.y - pm.t[1], P.z - pm.t[2]};
                mat33f_vec(pm.Ri, X, X);
                const float e0 = kp.at(3)[i] - q.x, e1 = kp.at(4)[i] - q.y;
                if (!(((double)fabsf(e0) > 3.0) || ((double)fabsf(e1) > 3.0))) {
                    pose_jacobian(fx, fy, X[0], X[1], X[2], J);
                    d0 = e0; d1 = e1;
                }
            }
            // a keypoint that does not take part stages zeros: 0*0 + 0*0 added to a sum changes nothing
#pragma unroll
            for (int k = 0; k < 12; k++) ((SVO_LDS(float)*)sh.js[k])[j] = J[k];
            ((SVO_LDS(float)*)sh.js[12])[j] = d0;
            ((SVO_LDS(float)*)sh.js[13])[j] = d1;
        }
        rp_sync<WAVES>();
        if (tid < 27) {
            const int m = min(RP_CHUNK, n - c0);
            const SVO_LDS(rp_v4f)* p0 = (const SVO_LDS(rp_v4f)*)sh.js[i0];
            const SVO_LDS(rp_v4f)* p1 = (const SVO_LDS(rp_v4f)*)sh.js[i1];
            const SVO_LDS(rp_v4f)* p2 = (const SVO_LDS(rp_v4f)*)sh.js[i2];
            const SVO_LDS(rp_v4f)* p3 = (const SVO_LDS(rp_v4f)*)sh.js[i3];
#pragma unroll 4
            for (int j = 0; j < (m + 3) >> 2; j++) {
                const rp_v4f x0 = p0[j], x1 = p1[j], x2 = p2[j], x3 = p3[j];
                { float s = 0; s += x0.x * x2.x; s += x1.x * x3.x; acc += s; }
                { float s = 0; s += x0.y * x2.y; s += x1.y * x3.y; acc += s; }
                { float s = 0; s += x0.z * x2.z; s += x1.z * x3.z; acc += s; }
                { float s = 0; s += x0.w * x2.w; s += x1.w * x3.w; acc += s; }
            }
        }
    }
    rp_sync<WAVES>();
    if (tid < 27) ((SVO_LDS(float)*)sh.sums)[tid] = acc;
    rp_sync<WAVES>();
    float H[36], e[6], twist[6];
    {
        int idx = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = r; c < 6; c++) {
                const float v = ((const SVO_LDS(float)*)sh.sums)[idx++];
                H[r * 6 + c] = v; H[c * 6 + r] = v;   // J[r]*J[c] == J[c]*J[r]: the lower triangle has the same bits
            }
#pragma unroll
        for (int r = 0; r < 6; r++) e[r] = ((const SVO_LDS(float)*)sh.sums)[21 + r];
    }
#ifdef SVO_SVD_ONE_LANE
    // experiment: the solve under an exec mask of one lane, result broadcast
    if (tid == 0) gn_solve6(H, e, twist, a.exact_pinv != 0);
#pragma unroll
    for (int q = 0; q < 6; q++) twist[q] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(twist[q])));
#else
    gn_solve6(H, e, twist, a.exact_pinv != 0);
#endif
    exponential_map(twist, grad);                  // not rotated (pose_refinement.cpp:398-411)
}

// Registers are capped at 512 / SVO_RP_OCC per lane. 2: the one-wave shape wanted 259 (5 go to scratch), and a
// wavefront that holds 264 of a SIMD's 512 registers leaves room for one KLT wavefront beside it, one that holds
// 256 for two (+0.4 % frames/s on top of the smaller KLT tile, profiles/r03_ab_steps.txt).
#ifndef SVO_RP_OCC
#define SVO_RP_OCC 2
#endif
#define RP_OCC_ATTR __attribute__((amdgpu_waves_per_eu(SVO_RP_OCC)))
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) RP_OCC_ATTR void reproj_gn_kernel(const ReprojArgs* __restrict__ args, int cap) {
    constexpr int T = 64 * WAVES;
    __builtin_amdgcn_s_setprio(3);     // latency-bound: win the issue arbitration against co-resident window kernels
    const ReprojArgs& a = args[blockIdx.x];
    const int n = min(*G(a.n_ptr), cap);
    const int tid = threadIdx.x;
    __shared__ ReprojShared<T> sh;
    extern __shared__ __attribute__((aligned(16))) float rp_dyn[];
    const RpKps kp{(SVO_LDS(float)*)rp_dyn, cap};

    if (a.tracked) {   // merge, pose_refinement.cpp:125-150
        for (int i = tid; i < n; i += T) {
            const svo_kp2d k = G(a.kps2d)[i], t = G(a.tracked)[i];
            const float dx = k.x - t.x, dy = k.y - t.y;
            const float diff = dx * dx + dy * dy;
            uint32_t f = G(a.flags)[i];
            if (G(a.err)[i] > 20) f |= SVO_IGNORE_COMPLETELY;
            else if (diff > 81) f |= SVO_IGNORE_DURING_REFINEMENT;
            else { f &= ~(uint32_t)SVO_IGNORE_DURING_REFINEMENT; G(a.kps2d)[i] = t; }
            G(a.flags)[i] = f;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += T) {     // the same lane that merged keypoint i
        const svo_kp3d P = G(a.kps3d)[i];
        const svo_kp2d k = G(a.kps2d)[i];
        const uint32_t f = G(a.flags)[i];
        kp.at(0)[i] = P.x; kp.at(1)[i] = P.y; kp.at(2)[i] = P.z;
        kp.at(3)[i] = k.x; kp.at(4)[i] = k.y;
        kp.at(5)[i] = (f & (SVO_IGNORE_DURING_REFINEMENT | SVO_IGNORE_COMPLETELY | SVO_IGNORE_TEMPORARY)) ? 0.f : 1.f;
    }
    rp_sync<WAVES>();

    float x0[6];
#pragma unroll
    for (int j = 0; j < 6; j++) x0[j] = G(a.pose_in)[j];
    const int maxIter = 50;
    int n_grad = 0, n_cost = 1, accepted = 0, exit_small = 0, par = 0;
    float prev_cost = reproj_cost<WAVES>(a, kp, n, x0, sh, par);
    const float initial = prev_cost;
    for (int i = 0; i < maxIter; i++) {
        float g[6];
        reproj_gradient<WAVES>(a, kp, n, x0, sh, g);
        n_grad++;
        float k = 1.0f;
        for (; i < maxIter; i++) {
            float x[6];
#pragma unroll
            for (int j = 0; j < 6; j++) x[j] = x0[j] + k * g[j];
            const float new_cost = reproj_cost<WAVES>(a, kp, n, x, sh, par);
            n_cost++;
            if (new_cost < prev_cost) {
#pragma unroll
                for (int j = 0; j < 6; j++) x0[j] = x[j];
                prev_cost = new_cost;
                accepted++;
                break;
            } else if ((double)fabsf(new_cost - prev_cost) < 0.0001) {
                i = maxIter;
                exit_small = 1;
                break;
            } else
                k /= 2;
        }
    }
    if (tid == 0) {
        for (int j = 0; j < 6; j++) G(a.pose_out)[j] = x0[j];
        if (a.cost_out) *G(a.cost_out) = prev_cost;
        if (a.zero_out) *G(a.zero_out) = 0;
        if (a.trace) {
            svo_gn_trace t;
            t.level = 0; t.n_gradient = n_grad; t.n_cost = n_cost; t.n_accepted = accepted;
            t.exit_small = exit_small; t.initial_cost = initial; t.final_cost = prev_cost;
            for (int j = 0; j < 6; j++) t.pose[j] = x0[j];
            *G(a.trace) = t;
        }
    }
}

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ pose, const svo_kp3d* __restrict__ kps3d,
                                                      int n, svo_camera_settings cam, svo_kp2d* __restrict__ out) {
    float p[6];
#pragma unroll
    for (int j = 0; j < 6; j++) p[j] = pose[j];
    PoseMats pm;
    pose_mats(p, pm);
    const CamD camd = make_camd(cam.fx, cam.fy, cam.cx, cam.cy, cam);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = project_point(pm.Rd, pm.t, camd, kps3d[i]);
}

void launch_project(const float* pose, const svo_kp3d* kps3d, int n, const svo_camera_settings& cam,
                    svo_kp2d* out, hipStream_t stream) {
    hipLaunchKernelGGL(project_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pose, kps3d, n, cam, out);
}

// n_bound: upper bound of the keypoint counts of the launch (sizes the LDS copy of the keypoints:
// 24 B each). One wave per sequence up to 128 keypoints, four beyond. Returns false when the
// keypoints do not fit LDS (more than ~5000).
bool launch_reproj(const ReprojArgs* d_args, int batch, int n_bound, hipStream_t stream) {
    // a batch of sequences: one wave each up to 256 keypoints (four staging steps): the four-wave shape took
    // 2.0 against 0.8 ms per launch in the round-3 profile, for every sequence of a launch whose largest set passed 128
    const int T = n_bound <= (batch >= 32 ? 256 : 128) ? 64 : 256;
    const int cap = (std::max(n_bound, 1) + T - 1) / T * T;          // whole staging steps
    const size_t lds = (size_t)cap * 6 * sizeof(float);
    if (lds > 120 * 1024) return false;
    static LdsLimit limit1, limit4;
    if (raise_lds_limit(limit1, reinterpret_cast<const void*>(reproj_gn_kernel<1>), 120 * 1024) != hipSuccess ||
        raise_lds_limit(limit4, reinterpret_cast<const void*>(reproj_gn_kernel<4>), 120 * 1024) != hipSuccess)
        return false;
    if (T == 64)
        hipLaunchKernelGGL(reproj_gn_kernel<1>, dim3(batch), dim3(64), lds, stream, d_args, cap);
    else
        hipLaunchKernelGGL(reproj_gn_kernel<4>, dim3(batch), dim3(256), lds, stream, d_args, cap);
    return true;
}

}  // namespace svo
