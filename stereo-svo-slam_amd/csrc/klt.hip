// klt.hip — row B2 of SURVEY §8a: OpticalFlow::calculate_optical_flow
// (src/lib/optical_flow.cpp:14-56) = cv::calcOpticalFlowPyrLK with a (w,w)
// window, maxLevel 2, 30 iterations / eps 0.01, OPTFLOW_USE_INITIAL_FLOW,
// tracking every keypoint from the Gaussian pyramid of its ORIGIN KEYFRAME
// into the current frame (PoseRefiner::refine_pose, src/lib/pose_refinement.cpp:72-118).
//
// One 128-thread workgroup (two wavefronts) per keypoint, all pyramid levels and
// iterations inside the kernel. Per level the (w+3)^2 neighbourhood of the reference point is staged
// in LDS with BORDER_REFLECT_101 addressing, the Scharr derivatives of
// OpenCV's pyramid are computed from that tile (zero outside the image, as
// the constant border of cv::buildOpticalFlowPyramid), and the fixed-point
// template (14-bit weights, 5 fractional bits) stays in LDS. The search image
// is staged once per level as a tile with a 6 px margin around the start
// position, so an iteration touches HBM only when the window drifts out of
// it. Lanes map to (row parity, column) of the window, so there is no
// integer division in the loops; all window sums are exact integers reduced
// with DPP row adds, so the result does not depend on the reduction order.
#include "svo_kernels.hpp"

namespace svo {

constexpr int KLT_MAX_WIN = 35;
constexpr int KLT_RW = KLT_MAX_WIN + 3;      // reference tile edge (window + 1 tap + 2 Scharr)
constexpr int KLT_DW = KLT_MAX_WIN + 1;      // derivative / tap grid edge
constexpr int KLT_MARGIN = 6;
constexpr int KLT_TJ = KLT_DW + 2 * KLT_MARGIN;   // search tile edge
#ifndef SVO_KLT_THREADS
#define SVO_KLT_THREADS 128
#endif
constexpr int KLT_THREADS = SVO_KLT_THREADS;
constexpr int KLT_WAVES = KLT_THREADS / 64;

// exact 64-bit sum of per-thread int32 partials over the workgroup
// (DPP row adds per wave, waves combined through LDS), same value in every thread
template <int NV>
__device__ inline void klt_block_sum(const int (&v)[NV], long long (&out)[NV], long long (*s_part)[4]) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const long long w = wave_sum_i32_to_i64(v[k]);
        if ((threadIdx.x & 63) == 0) s_part[wave][k] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        long long t = 0;
#pragma unroll
        for (int w = 0; w < KLT_WAVES; w++) t += s_part[w][k];
        out[k] = t;
    }
    __syncthreads();
}

#define SVO_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
// every product in the window loops has operands below 2^23 in magnitude (pixels < 2^8,
// weights <= 2^14, derivatives < 2^13, differences < 2^14): full-rate 24-bit multiplies
// instead of the quarter-rate 32-bit v_mul_lo_u32
#define M24(a, b) __mul24((a), (b))

__device__ inline int cv_round(float v) { return (int)rintf(v); }
__device__ inline int cv_floor(float v) { return (int)floorf(v); }

__global__ __launch_bounds__(KLT_THREADS) void klt_track_kernel(const KltArgs* __restrict__ args) {
    const KltArgs& a = args[blockIdx.y];
    const int n = *a.n_ptr;
    const int kp = blockIdx.x;
    if (kp >= n) return;
    const int lane = threadIdx.x;          // thread index in the workgroup
    const int win = a.win;
    const int RW = win + 3, DW = win + 1, TJ = DW + 2 * KLT_MARGIN;
    // thread -> (row offset, column): a row takes half a wave when it fits, else a whole wave
    const bool two = DW <= 32;
    const int lc = two ? (lane & 31) : (lane & 63);
    const int lr = two ? (lane >> 5) : (lane >> 6);
    const int rstep = two ? KLT_THREADS / 32 : KLT_THREADS / 64;
    // tile loads: one thread per column, the rows split over the threads sharing that column
    const int tcol = lane & 63, trow0 = lane >> 6;

    __shared__ uint8_t s_I[KLT_RW * KLT_RW];
    __shared__ int s_d[KLT_DW * KLT_DW];          // packed (dx, dy) int16
    __shared__ short s_Iw[KLT_MAX_WIN * KLT_MAX_WIN];
    __shared__ int s_dIw[KLT_MAX_WIN * KLT_MAX_WIN];
    __shared__ uint8_t s_J[KLT_TJ * KLT_TJ];
    __shared__ long long s_part[KLT_WAVES][4];

    const int kfid = a.kf_id ? a.kf_id[kp] : 0;
    const KfDev& kf = a.kfs[kfid];

    svo_kp2d ref;
    float nx, ny;
    if (a.proj_pose) {
        // frame.kps.kps2d = project_keypoints(estimated_pose, kps3d)   (stereo_slam.cpp:73-80)
        float pose[6];
        for (int i = 0; i < 6; i++) pose[i] = a.proj_pose[i];
        PoseMats pm;
        pose_mats(pose, pm);
        const CamD camd = make_camd(a.cam.fx, a.cam.fy, a.cam.cx, a.cam.cy, a.cam);
        const svo_kp2d q = project_point(pm.Rd, pm.t, camd, a.kps3d[kp]);
        nx = q.x; ny = q.y;
        ref = kf.kps2d[a.kp_index[kp]];
        if (lane == 0) {
            a.proj_out[kp] = q;
            if (a.ref_out) a.ref_out[kp] = ref;
        }
    } else {
        ref = a.prev_pts[kp];
        const svo_kp2d q = a.cur_pts[kp];
        nx = q.x; ny = q.y;
    }

    const int maxLevel = min(kf.n_lk, a.n_cur) - 1;
    const float halfWin = (win - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    double epsilon = 0.01;
    epsilon *= epsilon;
    int status = 1;
    float err = 0;

    for (int level = maxLevel; level >= 0; level--) {
        const ImgView I = kf.lk[level];
        const ImgView J = a.cur[level];
        const float lscale = (float)(1. / (1 << level));
        float prevx = ref.x * lscale, prevy = ref.y * lscale;
        if (level == maxLevel) { nx = nx * lscale; ny = ny * lscale; }
        else { nx = nx * 2.f; ny = ny * 2.f; }
        float nextx = nx, nexty = ny;

        prevx -= halfWin; prevy -= halfWin;
        const int iprevx = cv_floor(prevx), iprevy = cv_floor(prevy);
        if (iprevx < -win || iprevx >= I.w || iprevy < -win || iprevy >= I.h) {
            if (level == 0) { status = 0; err = 0; }
            continue;
        }
        float fa = prevx - iprevx, fb = prevy - iprevy;
        int iw00 = cv_round((1.f - fa) * (1.f - fb) * (1 << W_BITS));
        int iw01 = cv_round(fa * (1.f - fb) * (1 << W_BITS));
        int iw10 = cv_round((1.f - fa) * fb * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        __syncthreads();
        // (w+3)^2 tile of I around the window, rows iprevy-1 .., reflect-101
        if (tcol < RW) {
            const int gx = reflect101(iprevx - 1 + tcol, I.w);
            for (int r0 = trow0 * 8; r0 < RW; r0 += 8 * KLT_WAVES) {   // 8 independent loads in flight
                uint8_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int gy = reflect101(iprevy - 1 + min(r0 + u, RW - 1), I.h);
                    v[u] = I.data[M24(gy, I.stride) + gx];   // images are < 2^24 x 2^24, offsets < 2^31
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (r0 + u < RW) s_I[(r0 + u) * KLT_RW + tcol] = v[u];
            }
        }
        __syncthreads();
        // Scharr (calcSharrDeriv) at the (w+1)^2 tap positions; 0 outside the image
        if (lc < DW) {
            const int gx = iprevx + lc;
            const bool xin = (unsigned)gx < (unsigned)I.w;
            for (int r = lr; r < DW; r += rstep) {
                const int gy = iprevy + r;
                int packed = 0;
                if (xin && (unsigned)gy < (unsigned)I.h) {
                    const uint8_t* p0 = &s_I[r * KLT_RW + lc];      // row gy-1, col gx-1
                    const uint8_t* p1 = p0 + KLT_RW;
                    const uint8_t* p2 = p1 + KLT_RW;
                    const int t0m = M24(p0[0] + p2[0], 3) + M24(p1[0], 10), t0p = M24(p0[2] + p2[2], 3) + M24(p1[2], 10);
                    const int t1m = p2[0] - p0[0], t1c = p2[1] - p0[1], t1p = p2[2] - p0[2];
                    const int dx = t0p - t0m;
                    const int dy = M24(t1p + t1m, 3) + M24(t1c, 10);
                    packed = (dx & 0xffff) | (dy << 16);
                }
                s_d[r * KLT_DW + lc] = packed;
            }
        }
        __syncthreads();
        // template + covariance of derivatives (per-lane int32 partials: <= 18 pixels each)
        int a11 = 0, a12 = 0, a22 = 0;
        if (lc < win) {
            for (int y = lr; y < win; y += rstep) {
                const uint8_t* src = &s_I[(y + 1) * KLT_RW + lc + 1];
                const int* ds = &s_d[y * KLT_DW + lc];
                const int ival = SVO_DESCALE(M24(src[0], iw00) + M24(src[1], iw01) + M24(src[KLT_RW], iw10) +
                                             M24(src[KLT_RW + 1], iw11), W_BITS - 5);
                const int d00 = ds[0], d01 = ds[1], d10 = ds[KLT_DW], d11 = ds[KLT_DW + 1];
                const int ixval = SVO_DESCALE(M24((int)(short)d00, iw00) + M24((int)(short)d01, iw01) +
                                              M24((int)(short)d10, iw10) + M24((int)(short)d11, iw11), W_BITS);
                const int iyval = SVO_DESCALE(M24(d00 >> 16, iw00) + M24(d01 >> 16, iw01) +
                                              M24(d10 >> 16, iw10) + M24(d11 >> 16, iw11), W_BITS);
                s_Iw[y * KLT_MAX_WIN + lc] = (short)ival;
                s_dIw[y * KLT_MAX_WIN + lc] = (ixval & 0xffff) | (iyval << 16);
                a11 += M24(ixval, ixval); a12 += M24(ixval, iyval); a22 += M24(iyval, iyval);
            }
        }
        long long sA[3];
        {
            const int pa[3] = {a11, a12, a22};
            klt_block_sum<3>(pa, sA, s_part);
        }
        const long long iA11 = sA[0], iA12 = sA[1], iA22 = sA[2];
        const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE,
                    A22 = (float)iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (2 * win * win);
        if ((double)minEig < 1e-4 || D < FLT_EPSILON) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        nextx -= halfWin; nexty -= halfWin;
        float prevDx = 0, prevDy = 0;
        int tx0 = 0, ty0 = 0;
        bool have_tile = false;

        // stage the search tile [tx0, tx0+TJ) x [ty0, ty0+TJ) around (cx, cy)
        auto load_tile = [&](int cx, int cy) {
            tx0 = cx - KLT_MARGIN; ty0 = cy - KLT_MARGIN;
            __syncthreads();
            if (tcol < TJ) {
                const int gx = reflect101(tx0 + tcol, J.w);
                for (int r0 = trow0 * 8; r0 < TJ; r0 += 8 * KLT_WAVES) {
                    uint8_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int gy = reflect101(ty0 + min(r0 + u, TJ - 1), J.h);
                        v[u] = J.data[M24(gy, J.stride) + gx];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (r0 + u < TJ) s_J[(r0 + u) * KLT_TJ + tcol] = v[u];
                }
            }
            __syncthreads();
            have_tile = true;
        };

        for (int j = 0; j < 30; j++) {
            const int inextx = cv_floor(nextx), inexty = cv_floor(nexty);
            if (inextx < -win || inextx >= J.w || inexty < -win || inexty >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            if (!have_tile || inextx < tx0 || inexty < ty0 || inextx + DW > tx0 + TJ || inexty + DW > ty0 + TJ)
                load_tile(inextx, inexty);
            fa = nextx - inextx; fb = nexty - inexty;
            iw00 = cv_round((1.f - fa) * (1.f - fb) * (1 << W_BITS));
            iw01 = cv_round(fa * (1.f - fb) * (1 << W_BITS));
            iw10 = cv_round((1.f - fa) * fb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int b1 = 0, b2 = 0;
            if (lc < win) {
                const uint8_t* jbase = &s_J[(inexty - ty0) * KLT_TJ + (inextx - tx0) + lc];
                for (int y = lr; y < win; y += rstep) {
                    const uint8_t* jp = jbase + y * KLT_TJ;
                    const int diff = SVO_DESCALE(M24(jp[0], iw00) + M24(jp[1], iw01) + M24(jp[KLT_TJ], iw10) +
                                                 M24(jp[KLT_TJ + 1], iw11), W_BITS - 5) -
                                     (int)s_Iw[y * KLT_MAX_WIN + lc];
                    const int dI = s_dIw[y * KLT_MAX_WIN + lc];
                    b1 += M24(diff, (int)(short)dI);
                    b2 += M24(diff, dI >> 16);
                }
            }
            long long sB[2];
            {
                const int pb[2] = {b1, b2};
                klt_block_sum<2>(pb, sB, s_part);
            }
            const long long ib1 = sB[0], ib2 = sB[1];
            const float fb1 = (float)ib1 * FLT_SCALE, fb2 = (float)ib2 * FLT_SCALE;
            const float dx = (float)((A12 * fb2 - A22 * fb1) * D);
            const float dy = (float)((A12 * fb1 - A11 * fb2) * D);
            nextx += dx; nexty += dy;
            nx = nextx + halfWin; ny = nexty + halfWin;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && (double)fabsf(dx + prevDx) < 0.01 && (double)fabsf(dy + prevDy) < 0.01) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            prevDx = dx; prevDy = dy;
        }

        if (status && level == 0) {
            const float npx = nx - halfWin, npy = ny - halfWin;
            const int inx = cv_floor(npx), iny = cv_floor(npy);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                status = 0;
                continue;
            }
            if (!have_tile || inx < tx0 || iny < ty0 || inx + DW > tx0 + TJ || iny + DW > ty0 + TJ)
                load_tile(inx, iny);
            const float aa = npx - inx, bb = npy - iny;
            iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int e = 0;
            if (lc < win) {
                const uint8_t* jbase = &s_J[(iny - ty0) * KLT_TJ + (inx - tx0) + lc];
                for (int y = lr; y < win; y += rstep) {
                    const uint8_t* jp = jbase + y * KLT_TJ;
                    const int diff = SVO_DESCALE(M24(jp[0], iw00) + M24(jp[1], iw01) + M24(jp[KLT_TJ], iw10) +
                                                 M24(jp[KLT_TJ + 1], iw11), W_BITS - 5) -
                                     (int)s_Iw[y * KLT_MAX_WIN + lc];
                    e += diff < 0 ? -diff : diff;
                }
            }
            long long sE[1];
            {
                const int pe[1] = {e};
                klt_block_sum<1>(pe, sE, s_part);   // < 2^24: the float sum of |diff| is exact in any order
            }
            const float errval = (float)(int)sE[0];
            err = errval * 1.f / (32 * win * win);
        }
    }

    if (lane == 0) {
        a.cur_pts[kp] = svo_kp2d{nx, ny};
        a.status[kp] = (uint8_t)status;
        a.err[kp] = status ? err : INFINITY;   // optical_flow.cpp:46-50
    }
}

void launch_klt(const KltArgs* d_args, int batch, int max_n, int win, hipStream_t stream) {
    (void)win;
    if (max_n <= 0) return;
    hipLaunchKernelGGL(klt_track_kernel, dim3(max_n, batch), dim3(KLT_THREADS), 0, stream, d_args);
}

}  // namespace svo
