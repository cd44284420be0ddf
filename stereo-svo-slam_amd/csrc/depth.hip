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
    const uint8_t* src = im.g() + (size_t)y * im.stride + x;
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

// 16 consecutive pixels of row y starting at column x (any alignment), every valid byte ^ (flip & 0xff),
// bytes from `valid` on = (pad & 0xff): one 16-byte load when the row has them, dwords otherwise
__device__ inline uint4 load_u8x16(const ImgView& im, int y, int x, int valid, uint32_t flip, uint32_t pad) {
    uint32_t w[4];
    if (x + 16 <= im.w) {
        const uint8_t* src = im.g() + (size_t)y * im.stride + x;
        __builtin_memcpy(w, src, 16);                     // (one global_load_dwordx4: any byte alignment)
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = load_u8x4(im, y, x + 4 * k, 4);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int nv = valid - 4 * k;                     // valid bytes of this dword
        const uint32_t vm = nv >= 4 ? 0xFFFFFFFFu : (nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u));
        w[k] = ((w[k] ^ flip) & vm) | (pad & ~vm);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

constexpr int SSD_WIN_ALL = 35, SSD_MH_ALL = 17; // template edge and match-map rows (2*search_y+1) the entry points accept
constexpr int SSD_MAX_MW = 65;                  // match-map columns (search_x+1)
constexpr int SSD_T_STRIDE = 96;                // bytes per padded template row: 16 zeros | row | zeros
constexpr int SSD_R_STRIDE = 144;               // bytes per search-region row (36 dwords: conflict-free b128 rows)
constexpr int SSD_W_STRIDE = 105;               // ints per row of the column sums (= 1 mod 8: the 8 rows x 8 segments of 8 columns
                                                // that a wave reads in the row pass hit 64 different banks; 100 was a 4-way conflict)
#ifndef SVO_SSD_THREADS
#define SVO_SSD_THREADS 256
#endif
constexpr int SSD_THREADS = SVO_SSD_THREADS;

typedef int ssd_v4i __attribute__((ext_vector_type(4)));

// C1. One workgroup (4 waves) per keypoint. cv::matchTemplate(TM_SQDIFF) is
//   SSD[k][j] = sum_w R^2 - 2 sum_w R*T + sum T^2   over the window w at offset (k, j),
// exact in integers, and it is invariant to subtracting 128 from every pixel, so both images
// are staged as signed bytes (x ^ 0x80).
//  * The cross term is a correlation: 0.76 M multiply-adds per keypoint at the C2 settings,
//    the one place of this path with GEMM-class arithmetic. It runs on the matrix cores as a
//    Toeplitz product, one v_mfma_i32_16x16x64_i8 per template row r and 16-column block j0:
//        D[k][n] += sum_x A[k][x] * B[x][n],  A[k][x] = R[k + r][j0 + x]  (aligned 16 B / lane),
//                                             B[x][n] = T[r][x - n]       (0 outside the row),
//    so D[k][n] accumulates sum_{r,c} R[k+r][j0+n+c] T[r][c] over the th rows. Rows k >= mh and
//    columns j >= mw of D are computed and dropped. A wave owns a (16-row, 16-column) block of
//    the match map; B is rebuilt per row from the zero-padded template (5 dwords + v_alignbyte).
//    Any k-order inside the instruction is fine as long as A and B agree: lane l holds bytes
//    16*(l>>4) .. +15 of its row / column for both operands.
//  * sum_w R^2: sliding window sums, columns first (one thread per region column), then rows.
//  * argmin (first minimum in row-major order) and the tie-averaged column of :313-323 as before.
// SSD_MAX_WIN / SSD_MAX_MH: what the LDS is sized for. The window kernels of a step share each CU's 160 KB with
// the alignment kernel's 38 KB per wavefront, and what fits is what runs: the shape for windows up to 31 and
// search_y up to 6 (every configuration of the reference) takes 14.9 KB instead of 18.2.
template <int SSD_MAX_WIN, int SSD_MAX_MH>
__global__ __launch_bounds__(SSD_THREADS) void ssd_disparity_kernel(const SsdArgs* __restrict__ args) {
    constexpr int SSD_R_ROWS = SSD_MAX_WIN + SSD_MAX_MH, SSD_MAX_MATCH = SSD_MAX_MW * SSD_MAX_MH;
    const SsdArgs& a = args[blockIdx.y];
    if (a.enable && !*G(a.enable)) return;
    const int n = *G(a.n_ptr);
    const int kp = a.first + (a.first_ptr ? *G(a.first_ptr) : 0) + (int)blockIdx.x;
    if (kp >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    __shared__ __attribute__((aligned(16))) uint8_t s_t[(SSD_MAX_WIN + 1) * SSD_T_STRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t s_r[SSD_R_ROWS * SSD_R_STRIDE];
    __shared__ int s_w[SSD_MAX_MH * SSD_W_STRIDE];
    __shared__ int s_tt[4];
    __shared__ unsigned long long s_key[SSD_THREADS / 64];
    __shared__ int s_sum[SSD_THREADS / 64], s_cnt[SSD_THREADS / 64];

    const int window_before = a.win / 2, window_after = (a.win + 1) / 2;
    const int cols = a.left.w, rows = a.left.h;
    const svo_kp2d p = G(a.kps2d)[kp];
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
    if (tw > SSD_MAX_WIN || th > SSD_MAX_WIN || mw > SSD_MAX_MW || mh > SSD_MAX_MH || rw > SSD_W_STRIDE ||
        rh > SSD_R_ROWS)
        skip = true;  // the host validates window sizes; never taken with valid settings
    if (skip) {
        if (tid == 0) G(a.disparity)[kp] = -1.0f;
        return;
    }

    // ---- stage: template rows as [16 zero bytes | row ^ 0x80 | zeros], search region ^ 0x80 (bytes past
    // the region's width: 0x80), 16 bytes per lane and step: a template row is three chunks, a region row
    // up to six (round 2 staged dwords: 7 + 3 trips of ~22 instructions per thread against 2 + 1 here)
    static_assert(SSD_T_STRIDE % 16 == 0 && SSD_R_STRIDE % 16 == 0, "rows are whole 16-byte chunks");
    for (int i = tid; i < (SSD_MAX_WIN + 1) * (SSD_T_STRIDE / 16); i += SSD_THREADS) {
        const int r = i / (SSD_T_STRIDE / 16), ch = i % (SSD_T_STRIDE / 16);
        const int c = 16 * (ch - 1);                      // template column of the chunk's first byte
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < th && c >= 0 && c < tw) v = load_u8x16(a.left, y11 + r, x11 + c, tw - c, 0x80808080u, 0u);
        reinterpret_cast<uint4*>(s_t)[i] = v;
    }
    for (int i = tid; i < SSD_R_ROWS * (SSD_R_STRIDE / 16); i += SSD_THREADS) {
        const int r = i / (SSD_R_STRIDE / 16), ch = i % (SSD_R_STRIDE / 16);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < rh && 16 * ch < rw) v = load_u8x16(a.right, y21 + r, x21 + 16 * ch, rw - 16 * ch, 0x80808080u, 0x80808080u);
        reinterpret_cast<uint4*>(s_r)[i] = v;
    }
    __syncthreads();

    // ---- sum T^2 (wave 3) and the column sums W[k][x] = sum_{r<th} R[k+r][x]^2 (one thread per column)
    if (wave == SSD_THREADS / 64 - 1) {
        int tt = 0;
        if (lane < th) {
            const uint32_t* trow = reinterpret_cast<const uint32_t*>(s_t + lane * SSD_T_STRIDE) + 4;
#pragma unroll
            for (int d = 0; d < 9; d++) tt = __builtin_amdgcn_sdot4((int)trow[d], (int)trow[d], tt, false);
        }
        tt = wave_sum_dpp_i(tt);
        if (lane == 0) s_tt[0] = tt;
    }
    if (tid < rw) {
        const int8_t* col = reinterpret_cast<const int8_t*>(s_r) + tid;
        int sq = 0;
        for (int r = 0; r < th; r++) {
            const int v = col[r * SSD_R_STRIDE];
            sq += __mul24(v, v);
        }
        s_w[tid] = sq;
        for (int k = 1; k < mh; k++) {
            const int vn = col[(k + th - 1) * SSD_R_STRIDE], vo = col[(k - 1) * SSD_R_STRIDE];
            sq += __mul24(vn, vn) - __mul24(vo, vo);
            s_w[k * SSD_W_STRIDE + tid] = sq;
        }
    }

    // ---- cross term on the matrix cores: wave -> (row block, column block) of the match map
    const int n_jb = (mw + 15) >> 4, n_mb = (mh + 15) >> 4;
    const int fm = lane & 15, fh = lane >> 4;        // fragment row / column and 16-byte k slice
    constexpr int SSD_SLOTS = (10 + SSD_THREADS / 64 - 1) / (SSD_THREADS / 64);   // <= 2 x 5 blocks over the waves
    ssd_v4i acc[SSD_SLOTS];
    int task_of[SSD_SLOTS];
#pragma unroll
    for (int t = 0; t < SSD_SLOTS; t++) { acc[t] = ssd_v4i{0, 0, 0, 0}; task_of[t] = -1; }
    {
        int slot = 0;
        for (int task = wave; task < n_jb * n_mb && slot < SSD_SLOTS; task += SSD_THREADS / 64, slot++) {
            task_of[slot] = task;
            const int mb = task >= n_jb ? 1 : 0, jb = task - mb * n_jb;      // (at most two row blocks: the map has <= 17 rows)
            // B: bytes [16 + 16 fh - fm, +16) of the padded template row
            const int o = 16 + 16 * fh - fm;
            const uint32_t* tb = reinterpret_cast<const uint32_t*>(s_t) + (o >> 2);
            const unsigned sh = (unsigned)(o & 3);
            const uint8_t* ab = s_r + (16 * jb + 16 * fh);
            const int arow = 16 * mb + fm;
            ssd_v4i c = {0, 0, 0, 0};
            for (int r = 0; r < th; r++) {
                const uint32_t* t = tb + r * (SSD_T_STRIDE / 4);
                const uint32_t t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4];
                ssd_v4i bf;
                bf.x = (int)__builtin_amdgcn_alignbyte(t1, t0, sh);
                bf.y = (int)__builtin_amdgcn_alignbyte(t2, t1, sh);
                bf.z = (int)__builtin_amdgcn_alignbyte(t3, t2, sh);
                bf.w = (int)__builtin_amdgcn_alignbyte(t4, t3, sh);
                const ssd_v4i af = *reinterpret_cast<const ssd_v4i*>(ab + min(arow + r, SSD_R_ROWS - 1) * SSD_R_STRIDE);
                c = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, c, 0, 0, 0);
            }
            acc[slot] = c;
        }
    }
    __syncthreads();

    // ---- window sums of R^2: rows of W, 8 sliding segments per map row -> s_m. The match map takes
    // the place of the search region, which nobody reads after the barrier above (17.4 KB of LDS per
    // workgroup instead of 21.7: one more workgroup fits beside the alignment kernel's on a CU)
    static_assert(sizeof(int) * SSD_MAX_MATCH <= SSD_R_ROWS * SSD_R_STRIDE, "match map must fit the region tile");
    static_assert(SSD_MAX_MH <= 32, "two row blocks of 16 at most");
    int* const s_m = reinterpret_cast<int*>(s_r);
    {
        const int seg_len = (mw + 7) >> 3;
        for (int item = tid; item < mh * 8; item += SSD_THREADS) {
            const int k = item >> 3, j0 = (item & 7) * seg_len, j1 = min(mw, j0 + seg_len);
            if (j0 < j1) {
                const int* wrow = &s_w[k * SSD_W_STRIDE];
                int sq = 0;
                for (int c = 0; c < tw; c++) sq += wrow[j0 + c];
                s_m[k * mw + j0] = sq;
                for (int j = j0 + 1; j < j1; j++) {
                    sq += wrow[j + tw - 1] - wrow[j - 1];
                    s_m[k * mw + j] = sq;
                }
            }
        }
    }
    __syncthreads();
    {
        const int stt = s_tt[0];
#pragma unroll
        for (int slot = 0; slot < SSD_SLOTS; slot++) {
            const int task = task_of[slot];
            if (task < 0) continue;
            const int mb = task >= n_jb ? 1 : 0, jb = task - mb * n_jb;      // (at most two row blocks: the map has <= 17 rows)
            const int j = 16 * jb + fm;                     // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = 16 * mb + 4 * fh + i;
                if (k < mh && j < mw) s_m[k * mw + j] += stt - 2 * acc[slot][i];
            }
        }
    }
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
    // o / mw without an integer division: exact for o * mw < 2^20 (the map has at most 65 x 17 entries)
    const unsigned magic = ((1u << 20) + (unsigned)mw - 1u) / (unsigned)mw;
    for (int o = tid; o < nm; o += SSD_THREADS) {
        const int k = (int)(((unsigned)o * magic) >> 20), j = o - k * mw;
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
        G(a.disparity)[kp] = a.clamp_half ? fmaxf(0.5f, minPos) : minPos;
    }
}

void launch_ssd(const SsdArgs* d_args, int batch, int max_n, int win, int search_y, hipStream_t stream) {
    if (max_n <= 0) return;
    if (win <= 31 && 2 * search_y + 1 <= 13)
        hipLaunchKernelGGL((ssd_disparity_kernel<31, 13>), dim3(max_n, batch), dim3(SSD_THREADS), 0, stream, d_args);
    else
        hipLaunchKernelGGL((ssd_disparity_kernel<SSD_WIN_ALL, SSD_MH_ALL>), dim3(max_n, batch), dim3(SSD_THREADS), 0, stream, d_args);
}

// -------------------------------------------------------------------------
// One wavefront per 64 keypoints (grid: 64-keypoint blocks x sequences): single-wave workgroups find a
// slot as soon as any wave of the window kernels sharing the GPU retires, a 256-thread workgroup
// waits for four on one CU (measured 0.27 ms per launch against 0.013 ms alone). The frame's pose
// matrices are wave-uniform and computed by every lane; the inside counter is added atomically
// (the caller zeroes it: the reprojection kernel of the same frame, ReprojArgs::zero_out).
__global__ __launch_bounds__(64) void filter_update_kernel(const FilterArgs* __restrict__ args) {
    const FilterArgs& a = args[blockIdx.y];
    const int n = *G(a.n_ptr);
    if ((int)blockIdx.x * 64 >= n) return;
    const int tid = threadIdx.x;
    PoseMats s_frame;
    {
        float pose[6];
        for (int i = 0; i < 6; i++) pose[i] = G(a.frame_pose)[i];
        pose_mats(pose, s_frame);
    }
    const float fx = a.cam.fx, fy = a.cam.fy, cx = a.cam.cx, cy = a.cam.cy, baseline = a.cam.baseline;
    const CamD camd = make_camd(fx, fy, cx, cy, a.cam);
    int inside = 0;

    if (const int i = blockIdx.x * 64 + tid; i < n) {        // one keypoint per lane
        // references: explicit arrays (stage API) or the keyframe table (tracker)
        float kfp[6];
        svo_kp3d r3;
        svo_kp2d r2;
        SVO_GP(KfDev) kf = nullptr;
        int kidx = 0;
        if (a.kfs) {
            kf = &G(a.kfs)[G(a.kf_id)[i]];
            kidx = G(a.kp_index)[i];
            for (int q = 0; q < 6; q++) kfp[q] = kf->pose[q];
            r3 = G(kf->kps3d)[kidx];
            r2 = G(kf->kps2d)[kidx];
        } else {
            for (int q = 0; q < 6; q++) kfp[q] = G(a.kf_pose)[(size_t)i * 6 + q];
            r3 = svo_kp3d{0, 0, 0};
            r2 = svo_kp2d{0, 0};
            if (a.ref3d) r3 = G(a.ref3d)[i];
            if (a.ref2d) r2 = G(a.ref2d)[i];
        }
        PoseMats km;
        pose_mats(kfp, km);
        const svo_kp2d kp2 = G(a.kps2d)[i];
        uint32_t flags = G(a.flags)[i];
        int outl = G(a.outlier_count)[i], inl = G(a.inlier_count)[i];

        if (a.do_outlier_check) {   // depth_filter.cpp:52-128
            const float d = G(a.disparity)[i];
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

        svo_kp3d p3 = G(a.kps3d)[i];
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
                float kx = G(a.kf_inv_depth)[i], kP = G(a.kf_variance)[i];
                kf1_update(kx, kP, 0.0001f, Rm, 1 / _z);
                G(a.kf_inv_depth)[i] = kx;
                G(a.kf_variance)[i] = kP;
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
                G(kf->kps3d)[kidx] = p3;
                const uint32_t keep = G(kf->flags)[kidx] & SVO_IGNORE_DURING_REFINEMENT;
                G(kf->flags)[kidx] = keep | (flags & (SVO_IGNORE_TEMPORARY | SVO_IGNORE_COMPLETELY));
                G(kf->inlier_count)[kidx] = inl;
                G(kf->outlier_count)[kidx] = outl;
            }
        }
        G(a.kps3d)[i] = p3;
        if (a.do_flags) G(a.flags)[i] = flags;
        G(a.outlier_count)[i] = outl;
        G(a.inlier_count)[i] = inl;

        if (a.do_reproject) {       // stereo_slam.cpp:228-229 + keyframe_manager.cpp:55-64
            const svo_kp2d q = project_point(s_frame.Rd, s_frame.t, camd, p3);
            G(a.kps2d)[i] = q;
            if (q.x > 0 && q.y > 0 && q.x < a.width && q.y < a.height && !(flags & SVO_IGNORE_COMPLETELY))
                inside++;
        }
    }
    if (a.inside_count) {
        inside = wave_sum_i(inside);
        if (tid == 0 && inside) atomicAdd(a.inside_count, inside);
    }
}

void launch_filter(const FilterArgs* d_args, int batch, int max_n, hipStream_t stream) {
    if (max_n <= 0) return;
    hipLaunchKernelGGL(filter_update_kernel, dim3((max_n + 63) / 64, batch), dim3(64), 0, stream, d_args);
}

}  // namespace svo
