// depth.hip — rows C1, C2, D1, D3 of SURVEY §8a: the stereo depth filter
// DepthFilter::update_depth (src/lib/depth_filter.cpp:40-50).
//
//  * ssd_disparity_kernel (C1): calculate_disparities (depth_filter.cpp:259-327)
//    and the identical loop of DepthCalculator::calculate_depth
//    (depth_calculator.cpp:200-240). One workgroup per keypoint; template and
//    search region live in LDS; exact int32 SSD (cv::matchTemplate TM_SQDIFF),
//    first-minimum argmin in row-major order (cv::minMaxLoc) and the
//    tie-averaged column of :313-323.
//  * filter_update_kernel (C2, D1, D3): outlier_check (:52-128),
//    update_kps3d (:130-257), the flag/write-back loop of
//    StereoSlam::new_image (src/lib/stereo_slam.cpp:205-229) and the counter
//    of KeyFrameManager::keyframe_needed (keyframe_manager.cpp:47-74).
#include "svo_kernels.hpp"

namespace svo {

// 4 consecutive pixels of row y starting at column x as one (possibly unaligned)
// dword load, cut to `valid` bytes; byte loads only where the dword would run
// past the end of the image row
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ inline uint32_t load_u8x4(const ImgView& im, int y, int x, int valid) {
    const uint8_t* src = im.data + (size_t)y * im.stride + x;
    uint32_t v;
    if (x + 4 <= im.w) {
        v = *reinterpret_cast<const u32_unaligned*>(src);
    } else {
        v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (x + b < im.w) v |= (uint32_t)src[b] << (8 * b);
    }
    if (valid < 4) v &= (1u << (8 * valid)) - 1u;
    return v;
}

constexpr int SSD_MAX_WIN = 36;                 // template edge (window <= 35)
constexpr int SSD_NOFF = 4;                     // vertically adjacent offsets per work item
constexpr int SSD_T_STRIDE = 12;                // dwords (48 B: aligned 16 B reads)
constexpr int SSD_T_PAD = SSD_NOFF - 1;         // zero rows before the template
constexpr int SSD_T_ROWS = SSD_MAX_WIN + 2 * SSD_T_PAD + 4;
constexpr int SSD_R_STRIDE = 112;               // bytes per search-region row (>= 35+64+pad)
constexpr int SSD_R_ROWS = SSD_MAX_WIN + 16 + SSD_NOFF + 4;
constexpr int SSD_MAX_MATCH = 65 * 17;
constexpr int SSD_THREADS = 256;

// C1. One workgroup per keypoint. The template (zero padded to whole dwords,
// with zero rows above and below) and the search region sit in LDS. A work
// item is 4 vertically adjacent offsets (k0..k0+3, j): every search-region row
// is read once (TW4+1 dwords, funnel-shifted to the column phase j&3 with
// v_alignbyte), squared once (v_dot4_u32_u8 with itself) and multiplied
// against the 4 template rows it meets (v_dot4_u32_u8); the template rows
// slide through registers (one broadcast LDS read per step), so
// SSD = sum a^2 - 2 sum a*t + sum t^2 costs ~0.4 instructions per pixel pair
// and stays an exact integer. TW4 = dwords per template row (6, 8 or 9).
template <int TW4>
__device__ inline void ssd_load_trow(const uint32_t* s_t, int row, uint32_t (&t)[TW4]) {
    const uint4* t4 = reinterpret_cast<const uint4*>(&s_t[(row + SSD_T_PAD) * SSD_T_STRIDE]);
    uint32_t tv[12];
    *reinterpret_cast<uint4*>(&tv[0]) = t4[0];
    if (TW4 > 4) *reinterpret_cast<uint4*>(&tv[4]) = t4[1];
    if (TW4 > 8) *reinterpret_cast<uint4*>(&tv[8]) = t4[2];
#pragma unroll
    for (int d = 0; d < TW4; d++) t[d] = tv[d];
}

template <int TW4, bool ANYW>
__device__ inline void ssd_items(const uint32_t* s_t, const uint8_t* s_r, int* s_m, uint32_t stt,
                                 int tw, int th, int mw, int mh) {
    const int tid = threadIdx.x;
    // !ANYW: tw fills exactly TW4 dwords, only the last one is partial.
    // ANYW (windows clamped at the image border): a mask per dword.
    const int rem = tw - 4 * (TW4 - 1);
    const uint32_t last_mask = (rem >= 4 || rem <= 0) ? 0xFFFFFFFFu : ((1u << (8 * rem)) - 1u);
    uint32_t msk[TW4];
#pragma unroll
    for (int d = 0; d < TW4; d++) {
        const int rd = tw - 4 * d;
        msk[d] = rd >= 4 ? 0xFFFFFFFFu : (rd <= 0 ? 0u : ((1u << (8 * rd)) - 1u));
    }
    const int ngroups = (mh + SSD_NOFF - 1) / SSD_NOFF;
    const int nsteps = (th + SSD_NOFF - 1 + 3) & ~3;           // whole rotations of the 4 row registers
    for (int item = tid; item < ngroups * mw; item += SSD_THREADS) {
        const int kg = item / mw, j = item - kg * mw;
        const int k0 = kg * SSD_NOFF;
        const unsigned shift = (unsigned)(j & 3);
        const uint32_t* rbase = reinterpret_cast<const uint32_t*>(s_r) + (j >> 2) + k0 * (SSD_R_STRIDE / 4);
        uint32_t sab0 = 0, sab1 = 0, sab2 = 0, sab3 = 0, saa0 = 0, saa1 = 0, saa2 = 0, saa3 = 0;
        uint32_t ta[TW4], tb[TW4], tc[TW4], td[TW4];            // template rows st, st-1, st-2, st-3
#pragma unroll
        for (int d = 0; d < TW4; d++) { ta[d] = 0; tb[d] = 0; tc[d] = 0; td[d] = 0; }

        // one search-region row against the template rows (t0 newest .. t3 oldest)
        auto step = [&](int st, uint32_t (&t0)[TW4], const uint32_t (&t1)[TW4],
                        const uint32_t (&t2)[TW4], const uint32_t (&t3)[TW4]) {
            ssd_load_trow<TW4>(s_t, st, t0);
            const uint32_t* rr = rbase + st * (SSD_R_STRIDE / 4);
            uint32_t w[TW4 + 1];
#pragma unroll
            for (int d = 0; d <= TW4; d++) w[d] = rr[d];
            uint32_t rowsq = 0;
#pragma unroll
            for (int d = 0; d < TW4; d++) {
                uint32_t v = __builtin_amdgcn_alignbyte(w[d + 1], w[d], shift);   // column phase j & 3
                if (ANYW) v &= msk[d];
                else if (d == TW4 - 1) v &= last_mask;
                rowsq = __builtin_amdgcn_udot4(v, v, rowsq, false);
                sab0 = __builtin_amdgcn_udot4(v, t0[d], sab0, false);
                sab1 = __builtin_amdgcn_udot4(v, t1[d], sab1, false);
                sab2 = __builtin_amdgcn_udot4(v, t2[d], sab2, false);
                sab3 = __builtin_amdgcn_udot4(v, t3[d], sab3, false);
            }
            saa0 += (st < th) ? rowsq : 0u;
            saa1 += (st >= 1 && st - 1 < th) ? rowsq : 0u;
            saa2 += (st >= 2 && st - 2 < th) ? rowsq : 0u;
            saa3 += (st >= 3 && st - 3 < th) ? rowsq : 0u;
        };
        for (int st = 0; st < nsteps; st += 4) {
            step(st, ta, tb, tc, td);
            step(st + 1, td, ta, tb, tc);
            step(st + 2, tc, td, ta, tb);
            step(st + 3, tb, tc, td, ta);
        }
        if (k0 + 0 < mh) s_m[(k0 + 0) * mw + j] = (int)(saa0 + stt - 2u * sab0);
        if (k0 + 1 < mh) s_m[(k0 + 1) * mw + j] = (int)(saa1 + stt - 2u * sab1);
        if (k0 + 2 < mh) s_m[(k0 + 2) * mw + j] = (int)(saa2 + stt - 2u * sab2);
        if (k0 + 3 < mh) s_m[(k0 + 3) * mw + j] = (int)(saa3 + stt - 2u * sab3);
    }
}

__global__ __launch_bounds__(SSD_THREADS) void ssd_disparity_kernel(const SsdArgs* __restrict__ args) {
    const SsdArgs& a = args[blockIdx.y];
    if (a.enable && !*a.enable) return;
    const int n = *a.n_ptr;
    const int kp = a.first + (a.first_ptr ? *a.first_ptr : 0) + (int)blockIdx.x;
    if (kp >= n) return;
    const int tid = threadIdx.x;

    __shared__ __attribute__((aligned(16))) uint32_t s_t[SSD_T_ROWS * SSD_T_STRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t s_r[SSD_R_ROWS * SSD_R_STRIDE];
    __shared__ int s_m[SSD_MAX_MATCH];
    __shared__ uint32_t s_tt[SSD_MAX_WIN];
    __shared__ unsigned long long s_key[SSD_THREADS / 64];
    __shared__ int s_sum[SSD_THREADS / 64], s_cnt[SSD_THREADS / 64];

    const int window_before = a.win / 2, window_after = (a.win + 1) / 2;
    const int cols = a.left.w, rows = a.left.h;
    const svo_kp2d p = a.kps2d[kp];
    const int x = (int)p.x, y = (int)p.y;
    const int x11 = max(0, x - window_before);
    const int x12 = min(cols - 1, x + window_after);
    const int y11 = max(0, y - window_before);
    const int y12 = min(rows, y + window_after);
    const int x21 = x11;
    const int x22 = min(cols - 1, x + window_after + a.search_x);
    const int y21 = max(0, y - window_before - a.search_y);
    const int y22 = min(rows - 1, y + window_after + a.search_y);
    const int tw = x12 - x11, th = y12 - y11, rw = x22 - x21, rh = y22 - y21;
    const int mw = rw - tw + 1, mh = rh - th + 1;
    bool skip = false;
    if (a.clamp_half && (x12 <= 0 || y12 <= 0 || x11 >= cols - 1 || y11 >= rows - 1)) skip = true;
    if (a.clamp_half && (x22 <= 0 || y22 <= 0 || x21 >= cols - 1 || y21 >= rows - 1)) skip = true;
    if (tw <= 0 || th <= 0 || mw <= 0 || mh <= 0) skip = true;
    if (tw > SSD_MAX_WIN || th > SSD_MAX_WIN || rw > SSD_R_STRIDE - 12 || rh > SSD_R_ROWS - SSD_NOFF - 4 ||
        mw * mh > SSD_MAX_MATCH)
        skip = true;  // the host validates window sizes; never taken with valid settings
    if (skip) {
        if (tid == 0) a.disparity[kp] = -1.0f;
        return;
    }

    // ---- stage template (zero rows around it, zero beyond tw) and search region (zero outside)
    for (int i = tid; i < SSD_T_ROWS * SSD_T_STRIDE; i += SSD_THREADS) {
        const int r = i / SSD_T_STRIDE - SSD_T_PAD, d = i % SSD_T_STRIDE;
        uint32_t v = 0;
        if (r >= 0 && r < th && 4 * d < tw)
            v = load_u8x4(a.left, y11 + r, x11 + 4 * d, tw - 4 * d);
        s_t[i] = v;
    }
    for (int i = tid; i < SSD_R_ROWS * (SSD_R_STRIDE / 4); i += SSD_THREADS) {
        const int r = i / (SSD_R_STRIDE / 4), d = i % (SSD_R_STRIDE / 4);
        uint32_t v = 0;
        if (r < rh && 4 * d < rw)
            v = load_u8x4(a.right, y21 + r, x21 + 4 * d, rw - 4 * d);
        reinterpret_cast<uint32_t*>(s_r)[i] = v;
    }
    __syncthreads();
    if (tid < th) {
        uint32_t tt = 0;
#pragma unroll
        for (int d = 0; d < 9; d++) {
            const uint32_t t = s_t[(tid + SSD_T_PAD) * SSD_T_STRIDE + d];
            tt = __builtin_amdgcn_udot4(t, t, tt, false);
        }
        s_tt[tid] = tt;
    }
    __syncthreads();
    uint32_t stt = 0;
    for (int r = 0; r < th; r++) stt += s_tt[r];

    const int tw4 = (tw + 3) >> 2;
    if (tw4 == 8) ssd_items<8, false>(s_t, s_r, s_m, stt, tw, th, mw, mh);        // window 29..32
    else if (tw4 == 9) ssd_items<9, false>(s_t, s_r, s_m, stt, tw, th, mw, mh);   // window 33..36
    else if (tw4 == 6) ssd_items<6, false>(s_t, s_r, s_m, stt, tw, th, mw, mh);   // window 21..24
    else ssd_items<9, true>(s_t, s_r, s_m, stt, tw, th, mw, mh);                  // anything else
    __syncthreads();

    const int nm = mw * mh;
    unsigned long long best = ~0ull;
    for (int o = tid; o < nm; o += SSD_THREADS) {
        const unsigned long long key = ((unsigned long long)(unsigned)s_m[o] << 32) | (unsigned)o;
        best = key < best ? key : best;
    }
    // first minimum in row-major order == smallest (value, index) key
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other < best ? other : best;
    }
    if ((tid & 63) == 0) s_key[tid >> 6] = best;
    __syncthreads();
    best = s_key[0];
    for (int w = 1; w < SSD_THREADS / 64; w++) best = s_key[w] < best ? s_key[w] : best;
    const int min_int = (int)(best >> 32);
    const int min_o = (int)(best & 0xffffffffu);
    const int minx = min_o % mw, miny = min_o / mw;
    const float minVal = (float)min_int;

    int sumj = 0, cnt = 0;
    for (int o = tid; o < nm; o += SSD_THREADS) {
        const int k = o / mw, j = o % mw;
        if (j >= minx && k >= miny && (double)(float)s_m[o] <= (double)minVal) { sumj += j; cnt++; }
    }
    sumj = wave_sum_dpp_i(sumj);
    cnt = wave_sum_dpp_i(cnt);
    if ((tid & 63) == 0) { s_sum[tid >> 6] = sumj; s_cnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        int ts = 0, tc = 0;
        for (int w = 0; w < SSD_THREADS / 64; w++) { ts += s_sum[w]; tc += s_cnt[w]; }
        float minPos = (float)ts;      // sum of small ints: exact in float in any order
        minPos = minPos / tc;
        a.disparity[kp] = a.clamp_half ? fmaxf(0.5f, minPos) : minPos;
    }
}

void launch_ssd(const SsdArgs* d_args, int batch, int max_n, hipStream_t stream) {
    if (max_n <= 0) return;
    hipLaunchKernelGGL(ssd_disparity_kernel, dim3(max_n, batch), dim3(SSD_THREADS), 0, stream, d_args);
}

// -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void filter_update_kernel(const FilterArgs* __restrict__ args) {
    const FilterArgs& a = args[blockIdx.x];
    const int n = *a.n_ptr;
    const int tid = threadIdx.x;
    __shared__ PoseMats s_frame;
    __shared__ int s_inside[4];
    if (tid == 0) {
        float pose[6];
        for (int i = 0; i < 6; i++) pose[i] = a.frame_pose[i];
        pose_mats(pose, s_frame);
    }
    __syncthreads();
    const float fx = a.cam.fx, fy = a.cam.fy, cx = a.cam.cx, cy = a.cam.cy, baseline = a.cam.baseline;
    const CamD camd = make_camd(fx, fy, cx, cy, a.cam);
    int inside = 0;

    for (int i = tid; i < n; i += 256) {
        // references: explicit arrays (stage API) or the keyframe table (tracker)
        float kfp[6];
        svo_kp3d r3;
        svo_kp2d r2;
        KfDev* kf = nullptr;
        int kidx = 0;
        if (a.kfs) {
            kf = &a.kfs[a.kf_id[i]];
            kidx = a.kp_index[i];
            for (int q = 0; q < 6; q++) kfp[q] = kf->pose[q];
            r3 = kf->kps3d[kidx];
            r2 = kf->kps2d[kidx];
        } else {
            for (int q = 0; q < 6; q++) kfp[q] = a.kf_pose[(size_t)i * 6 + q];
            r3 = a.ref3d ? a.ref3d[i] : svo_kp3d{0, 0, 0};
            r2 = a.ref2d ? a.ref2d[i] : svo_kp2d{0, 0};
        }
        PoseMats km;
        pose_mats(kfp, km);
        const svo_kp2d kp2 = a.kps2d[i];
        uint32_t flags = a.flags[i];
        int outl = a.outlier_count[i], inl = a.inlier_count[i];

        if (a.do_outlier_check) {   // depth_filter.cpp:52-128
            const float d = a.disparity[i];
            const float _z = baseline / fmaxf(d, 0.5f);
            const float _x = (kp2.x - cx) / fx * _z;
            const float _y = (kp2.y - cy) / fy * _z;
            float pw[3] = {_x, _y, _z};
            mat33f_vec(s_frame.R, pw, pw);
            pw[0] += s_frame.t[0]; pw[1] += s_frame.t[1]; pw[2] += s_frame.t[2];
            float av[3] = {pw[0] - kfp[0], pw[1] - kfp[1], pw[2] - kfp[2]};
            mat33f_vec(km.Ri, av, av);
            float rv[3] = {r3.x - kfp[0], r3.y - kfp[1], r3.z - kfp[2]};
            mat33f_vec(km.Ri, rv, rv);
            const float disp_ref = baseline / rv[2];
            const float disp = baseline / av[2];
            const float pixel_distance = disp - disp_ref;
            if (fabsf(pixel_distance) > 5 * 0.5f) outl++;
            else inl++;
        }

        svo_kp3d p3 = a.kps3d[i];
        if (a.do_update) {          // depth_filter.cpp:130-257
            const float c1[3] = {kfp[0], kfp[1], kfp[2]};
            const float c2[3] = {s_frame.t[0], s_frame.t[1], s_frame.t[2]};
            float diff[3] = {fabsf(c1[0] - c2[0]), fabsf(c1[1] - c2[1]), fabsf(c1[2] - c2[2])};
            mat33f_vec(km.Ri, diff, diff);
            if (flags & (SVO_IGNORE_COMPLETELY | SVO_IGNORE_DURING_REFINEMENT)) {
                outl++;
            } else if (!((double)diff[0] < 0.1 && (double)diff[1] < 0.1)) {
                float p1[3] = {r2.x - cx, r2.y - cy, fx};
                mat33f_vec(km.R, p1, p1);
                float p2[3] = {kp2.x - cx, kp2.y - cy, fx};
                mat33f_vec(s_frame.R, p2, p2);
                const float A[6] = {p1[0], -p2[0], p1[1], -p2[1], p1[2], -p2[2]};
                const float yv[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
                float l[2];
                solve_svd_3x2(A, yv, l);
                const float deviation =
                    (float)(0.5 / (double)sqrtf(diff[0] * diff[0] + diff[1] * diff[1]));
                const float Rm = deviation * deviation;
                float Mx[9];
#pragma unroll
                for (int k = 0; k < 9; k++) Mx[k] = km.Ri[k] * l[0];
                const float pc[3] = {p1[0] - c1[0], p1[1] - c1[1], p1[2] - c1[2]};
                float new_p[3];
                mat33f_vec(Mx, pc, new_p);
                float _z = new_p[2];
                float kx = a.kf_inv_depth[i], kP = a.kf_variance[i];
                kf1_update(kx, kP, 0.0001f, Rm, 1 / _z);
                a.kf_inv_depth[i] = kx;
                a.kf_variance[i] = kP;
                _z = (float)(1.0 / (double)kx);
                const float _x = (r2.x - cx) / fx * _z;
                const float _y = (r2.y - cy) / fy * _z;
                float cp[3] = {_x, _y, _z};
                mat33f_vec(km.R, cp, cp);
                p3.x = c1[0] + cp[0];
                p3.y = c1[1] + cp[1];
                p3.z = c1[2] + cp[2];
            }
        }

        if (a.do_flags) {           // stereo_slam.cpp:205-226
            if (outl > inl) flags |= SVO_IGNORE_COMPLETELY;
            if (inl > outl) flags &= ~(uint32_t)SVO_IGNORE_TEMPORARY;
            if (kf) {
                kf->kps3d[kidx] = p3;
                const uint32_t keep = kf->flags[kidx] & SVO_IGNORE_DURING_REFINEMENT;
                kf->flags[kidx] = keep | (flags & (SVO_IGNORE_TEMPORARY | SVO_IGNORE_COMPLETELY));
                kf->inlier_count[kidx] = inl;
                kf->outlier_count[kidx] = outl;
            }
        }
        a.kps3d[i] = p3;
        if (a.do_flags) a.flags[i] = flags;
        a.outlier_count[i] = outl;
        a.inlier_count[i] = inl;

        if (a.do_reproject) {       // stereo_slam.cpp:228-229 + keyframe_manager.cpp:55-64
            const svo_kp2d q = project_point(s_frame.Rd, s_frame.t, camd, p3);
            a.kps2d[i] = q;
            if (q.x > 0 && q.y > 0 && q.x < a.width && q.y < a.height && !(flags & SVO_IGNORE_COMPLETELY))
                inside++;
        }
    }
    if (a.inside_count) {
        inside = wave_sum_i(inside);
        if ((tid & 63) == 0) s_inside[tid >> 6] = inside;
        __syncthreads();
        if (tid == 0) *a.inside_count = s_inside[0] + s_inside[1] + s_inside[2] + s_inside[3];
    }
}

void launch_filter(const FilterArgs* d_args, int batch, hipStream_t stream) {
    hipLaunchKernelGGL(filter_update_kernel, dim3(batch), dim3(256), 0, stream, d_args);
}

}  // namespace svo
