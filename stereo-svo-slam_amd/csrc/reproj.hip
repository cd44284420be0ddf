// reproj.hip — rows B1 and B3 of SURVEY §8a: the merge step of
// PoseRefiner::refine_pose (src/lib/pose_refinement.cpp:125-150) and the
// reprojection Gauss-Newton PoseRefiner::update_pose (:236-290) with
// PoseRefinerCallback::do_calc (:321-348) / get_gradient (:350-412).
//
// One persistent workgroup per sequence runs the whole line-search GN on the
// device: projection and residuals one thread per keypoint, the 21+6
// normal-equation sums by wavefront shuffles + one LDS pass, the 6x6
// pseudo-inverse, exponential map and the accept / halve / stop decisions on
// lane 0 — no host round trip per iteration.
#include "svo_kernels.hpp"
#include "svo_reduce.hpp"

namespace svo {

constexpr int RP_THREADS = 256;

struct ReprojShared {
    PoseMats pm;
    float red[RP_THREADS / 64][32];
    float sums[32];
    float grad[6];
};

// cost of PoseRefinerCallback::do_calc
__device__ float reproj_cost(const ReprojArgs& a, int n, const float pose[6], ReprojShared& sh) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) pose_mats(pose, sh.pm);
    __syncthreads();
    const CamD camd = make_camd(a.cam.fx, a.cam.fy, a.cam.cx, a.cam.cy, a.cam);
    float v[1] = {0};
    for (int i = tid; i < n; i += RP_THREADS) {
        const uint32_t f = a.flags[i];
        if (f & (SVO_IGNORE_DURING_REFINEMENT | SVO_IGNORE_COMPLETELY | SVO_IGNORE_TEMPORARY)) continue;
        const svo_kp2d q = project_point(sh.pm.Rd, sh.pm.t, camd, a.kps3d[i]);
        const svo_kp2d k = a.kps2d[i];
        const float d0 = fabsf(q.x - k.x), d1 = fabsf(q.y - k.y);
        v[0] += d0 + d1;
    }
    block_reduce<1, RP_THREADS>(v, sh.red, sh.sums);
    return sh.sums[0];
}

__device__ void reproj_gradient(const ReprojArgs& a, int n, const float pose[6], ReprojShared& sh) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) pose_mats(pose, sh.pm);
    __syncthreads();
    const CamD camd = make_camd(a.cam.fx, a.cam.fy, a.cam.cx, a.cam.cy, a.cam);
    const float fx = a.cam.fx, fy = a.cam.fy;
    float v[27];
#pragma unroll
    for (int k = 0; k < 27; k++) v[k] = 0;
    for (int i = tid; i < n; i += RP_THREADS) {
        const uint32_t f = a.flags[i];
        if (f & (SVO_IGNORE_DURING_REFINEMENT | SVO_IGNORE_COMPLETELY | SVO_IGNORE_TEMPORARY)) continue;
        const svo_kp3d P = a.kps3d[i];
        const svo_kp2d q = project_point(sh.pm.Rd, sh.pm.t, camd, P);
        float X[3] = {P.x - sh.pm.t[0], P.y - sh.pm.t[1], P.z - sh.pm.t[2]};
        mat33f_vec(sh.pm.Ri, X, X);
        float J[12];
        pose_jacobian(fx, fy, X[0], X[1], X[2], J);
        const svo_kp2d k = a.kps2d[i];
        const float d0 = k.x - q.x, d1 = k.y - q.y;
        if (((double)fabsf(d0) > 3.0) || ((double)fabsf(d1) > 3.0)) continue;
        int idx = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = r; c < 6; c++) {
                float s = 0;
                s += J[r] * J[c];
                s += J[6 + r] * J[6 + c];
                v[idx++] += s;
            }
#pragma unroll
        for (int r = 0; r < 6; r++) {
            float s = 0;
            s += J[r] * d0;
            s += J[6 + r] * d1;
            v[21 + r] += s;
        }
    }
    block_reduce<27, RP_THREADS>(v, sh.red, sh.sums);
    if (tid == 0) {
        float H[36], e[6], twist[6];
        int idx = 0;
        for (int r = 0; r < 6; r++)
            for (int c = r; c < 6; c++) { H[r * 6 + c] = sh.sums[idx]; H[c * 6 + r] = sh.sums[idx]; idx++; }
        for (int r = 0; r < 6; r++) e[r] = sh.sums[21 + r];
        gn_solve6(H, e, twist, a.exact_pinv != 0);
        exponential_map(twist, sh.grad);   // not rotated (pose_refinement.cpp:398-411)
    }
    __syncthreads();
}

__global__ __launch_bounds__(RP_THREADS) void reproj_gn_kernel(const ReprojArgs* __restrict__ args) {
    const ReprojArgs& a = args[blockIdx.x];
    const int n = *a.n_ptr;
    const int tid = threadIdx.x;
    __shared__ ReprojShared sh;

    if (a.tracked) {   // merge, pose_refinement.cpp:125-150
        for (int i = tid; i < n; i += RP_THREADS) {
            const svo_kp2d k = a.kps2d[i], t = a.tracked[i];
            const float dx = k.x - t.x, dy = k.y - t.y;
            const float diff = dx * dx + dy * dy;
            uint32_t f = a.flags[i];
            if (a.err[i] > 20) f |= SVO_IGNORE_COMPLETELY;
            else if (diff > 81) f |= SVO_IGNORE_DURING_REFINEMENT;
            else { f &= ~(uint32_t)SVO_IGNORE_DURING_REFINEMENT; a.kps2d[i] = t; }
            a.flags[i] = f;
        }
        __syncthreads();
    }

    float x0[6];
    for (int j = 0; j < 6; j++) x0[j] = a.pose_in[j];
    const int maxIter = 50;
    int n_grad = 0, n_cost = 1, accepted = 0, exit_small = 0;
    float prev_cost = reproj_cost(a, n, x0, sh);
    const float initial = prev_cost;
    for (int i = 0; i < maxIter; i++) {
        reproj_gradient(a, n, x0, sh);
        n_grad++;
        float g[6];
        for (int j = 0; j < 6; j++) g[j] = sh.grad[j];
        float k = 1.0f;
        for (; i < maxIter; i++) {
            float x[6];
            for (int j = 0; j < 6; j++) x[j] = x0[j] + k * g[j];
            const float new_cost = reproj_cost(a, n, x, sh);
            n_cost++;
            if (new_cost < prev_cost) {
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
        for (int j = 0; j < 6; j++) a.pose_out[j] = x0[j];
        if (a.cost_out) *a.cost_out = prev_cost;
        if (a.trace) {
            svo_gn_trace t;
            t.level = 0; t.n_gradient = n_grad; t.n_cost = n_cost; t.n_accepted = accepted;
            t.exit_small = exit_small; t.initial_cost = initial; t.final_cost = prev_cost;
            for (int j = 0; j < 6; j++) t.pose[j] = x0[j];
            *a.trace = t;
        }
    }
}

void launch_reproj(const ReprojArgs* d_args, int batch, hipStream_t stream) {
    hipLaunchKernelGGL(reproj_gn_kernel, dim3(batch), dim3(RP_THREADS), 0, stream, d_args);
}

}  // namespace svo
