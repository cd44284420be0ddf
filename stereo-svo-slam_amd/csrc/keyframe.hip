// keyframe.hip — the per-frame bookkeeping kernels of StereoSlam::new_image
// and keyframe creation, all on the device so that a frame never needs the
// image back on the host:
//
//  * compact_kernel: remove_outliers (src/lib/stereo_slam.cpp:43-56) and
//    find_bad_keypoints (src/lib/depth_calculator.cpp:67-86): order-preserving
//    stream compaction of the SoA keypoint set (workgroup prefix scan).
//  * kf_detect_kernel: CornerDetector::detect_keypoints
//    (src/lib/corner_detector.cpp:13-79) on every pyramid level at once:
//    cv::FAST(6, nonmax) restated as score + 3x3 NMS on an LDS tile per grid
//    cell, cv::Sobel(dx, 8U) edgelets as the fall-back, first-best selection in
//    the reference's scan orders.
//  * kf_select_merge_kernel: select_best_keypoints (:37-65, index-wise across
//    levels as the reference does) and merge_keypoints (:88-130, with the
//    swapped grid arguments of :179-180).
//  * kf_init_kernel: the depth / filter initialisation of
//    DepthCalculator::calculate_depth (:241-289), the copy into the new
//    keyframe (keyframe_manager.cpp:15-32) and the ignore_temporary rules of
//    stereo_slam.cpp:157-159 and :237-245.
#include "svo_kernels.hpp"
#include "svo_tracker.hpp"
#include <cstdlib>

namespace svo {

// ------------------------------------------------------------ prefix scan
// exclusive scan of one int per thread over the workgroup (any multiple of 64 threads)
__device__ inline int block_exclusive_scan(int v, int* s_wave /*[16]*/, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
        const int x = s_wave[w];
        if (w < wave) base += x;
        tot += x;
    }
    total = tot;
    return base + inc - v;
}

__device__ inline void copy_kp(const KpsDev& d, int di, const KpsDev& s, int si) {
    G(d.kps2d)[di] = G(s.kps2d)[si];
    G(d.kps3d)[di] = G(s.kps3d)[si];
    G(d.flags)[di] = G(s.flags)[si];
    G(d.kf_id)[di] = G(s.kf_id)[si];
    G(d.kp_index)[di] = G(s.kp_index)[si];
    G(d.outl)[di] = G(s.outl)[si];
    G(d.inl)[di] = G(s.inl)[si];
    G(d.kfx)[di] = G(s.kfx)[si];
    G(d.kfP)[di] = G(s.kfP)[si];
    G(d.score)[di] = G(s.score)[si];
    G(d.level_type)[di] = G(s.level_type)[si];
    G(d.color)[di] = G(s.color)[si];
}

__global__ __launch_bounds__(1024) void compact_kernel(const CompactArgs* __restrict__ args) {
    const CompactArgs& a = args[blockIdx.x];
    if (a.enable && !*G(a.enable)) return;
    __shared__ int s_wave[16];
    __shared__ int s_min_kf;
    __shared__ unsigned s_live[2];
    const int tid = threadIdx.x;
    if (tid == 0) { s_min_kf = INT_MAX; s_live[0] = s_live[1] = 0u; }   // (the scan's barriers stand between this and the atomics below)
    const int n = *G(a.src.n);
    const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
    const int i0 = tid * per, i1 = min(n, i0 + per);
    int cnt = 0;
    for (int i = i0; i < i1; i++) {
        const uint32_t f = G(a.src.flags)[i];
        bool keep;
        if (a.mode == 0) {
            keep = !(f & SVO_IGNORE_COMPLETELY);
        } else {
            const svo_kp2d k = G(a.src.kps2d)[i];
            keep = !((k.x < 0) || (k.y < 0) || (k.x > a.width) || (k.y > a.height) ||
                     (f & SVO_IGNORE_COMPLETELY) || (f & SVO_IGNORE_DURING_REFINEMENT));
        }
        cnt += keep ? 1 : 0;
    }
    int total;
    int pos = block_exclusive_scan(cnt, s_wave, total);
    int min_kf = INT_MAX;
    for (int i = i0; i < i1; i++) {
        const uint32_t f = G(a.src.flags)[i];
        bool keep;
        if (a.mode == 0) {
            keep = !(f & SVO_IGNORE_COMPLETELY);
        } else {
            const svo_kp2d k = G(a.src.kps2d)[i];
            keep = !((k.x < 0) || (k.y < 0) || (k.x > a.width) || (k.y > a.height) ||
                     (f & SVO_IGNORE_COMPLETELY) || (f & SVO_IGNORE_DURING_REFINEMENT));
        }
        if (keep) {
            copy_kp(a.dst, pos++, a.src, i);
            min_kf = min(min_kf, G(a.src.kf_id)[i]);
        }
    }
    if (tid == 0) *G(a.dst.n) = total;
    if (a.min_kf) {                            // (uniform) which keyframes the frame's keypoints still refer to
        if (min_kf != INT_MAX) atomicMin(&s_min_kf, min_kf);
        __syncthreads();
        const int base = s_min_kf;
        for (int i = i0; i < i1; i++) {
            if (G(a.src.flags)[i] & SVO_IGNORE_COMPLETELY) continue;      // (mode 0: exactly the keypoints kept above)
            const int j = G(a.src.kf_id)[i] - base;
            if (j >= 0 && j < 64) atomicOr(&s_live[j >> 5], 1u << (j & 31));
        }
        __syncthreads();
        if (tid == 0) { G(a.min_kf)[0] = base; G(a.min_kf)[1] = (int)s_live[0]; G(a.min_kf)[2] = (int)s_live[1]; }
    }
    if (a.zero)
        for (int i = tid; i < a.zero_count; i += blockDim.x) G(a.zero)[i] = 0;
}

// Small sets run with 256 threads: a 16-wave workgroup only starts on a CU that has drained, and
// next to the other sequence groups' window kernels it waited ~0.4 ms for one (HIP events).
void launch_compact(const CompactArgs* d_args, int batch, int cap, hipStream_t stream) {
    static const int env_threads = getenv("SVO_COMPACT_THREADS") ? atoi(getenv("SVO_COMPACT_THREADS")) : 0;   // (experiments)
    const int threads = env_threads >= 64 && env_threads <= 1024 && env_threads % 64 == 0 ? env_threads : (cap <= 1024 ? 256 : 1024);
    hipLaunchKernelGGL(compact_kernel, dim3(batch), dim3(threads), 0, stream, d_args);
}

// --------------------------------------------------------------- detection
constexpr int DET_MAX_CW = 96, DET_MAX_CH = 64;      // largest cell (grid_width x grid_height) the tracker accepts

__constant__ int c_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ int c_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// cv::FAST TYPE_9_16 (OpenCV fast.cpp / fast_score.cpp) in two steps: the corner test (9 contiguous ring
// pixels darker or brighter than the centre by more than the threshold) on every pixel, and
// cornerScore<16> on the corners only (kf_detect_kernel gathers them first: the score is 2/3 of the work
// and most pixels are no corners, but nearly every wavefront holds one)
__device__ inline bool fast_is_corner(const uint8_t* p, int stride, int threshold) {
    // ring pixel q is darker than the centre v by more than the threshold iff q - (v - t) < 0, brighter iff
    // (v + t) - q < 0: the sign bits are shifted into two 16-bit masks (v_alignbit: one instruction per bit),
    // and "nine contiguous on the circle" is a run of nine in the mask repeated twice
    const int v = p[0];
    const int lo = v - threshold, hi = v + threshold;
    unsigned md = 0, mb = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int q = p[c_ring_dy[k] * stride + c_ring_dx[k]];
        md = __builtin_amdgcn_alignbit(md, (unsigned)(q - lo), 31);
        mb = __builtin_amdgcn_alignbit(mb, (unsigned)(hi - q), 31);
    }
    auto run9 = [](unsigned m) {
        const unsigned w = m | (m << 16);
        unsigned r = w & (w >> 1);          // runs of 2, 4, 8 starting at a bit
        r &= r >> 2;
        r &= r >> 4;
        return (r & (w >> 8)) != 0;         // ... and the ninth
    };
    return run9(md) || run9(mb);
}
__device__ inline int fast_corner_score(const uint8_t* p, int stride, int threshold) {
    // cornerScore<16>: OpenCV leaves an arc as soon as its first three (five) differences cannot raise the
    // score; without those exits the result is the same (the running min / max of an arc only falls / rises,
    // so it ends at or below a0 / at or above b0), the code has no divergent branches, and min / max pair up
    // as v_min3 / v_max3. Pair minima are shared by the eight arcs: m(k) = min d[k+1 .. k+8].
    int d[25];
    const int v = p[0];
#pragma unroll
    for (int k = 0; k < 25; k++) d[k] = v - (int)p[c_ring_dy[k & 15] * stride + c_ring_dx[k & 15]];
    int pmin[12], pmax[12];                       // of d[2j+1], d[2j+2]
#pragma unroll
    for (int j = 0; j < 12; j++) { pmin[j] = min(d[2 * j + 1], d[2 * j + 2]); pmax[j] = max(d[2 * j + 1], d[2 * j + 2]); }
    int a0 = threshold;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const int j = k >> 1;
        const int a = min(min(min(pmin[j], pmin[j + 1]), pmin[j + 2]), pmin[j + 3]);
        a0 = max(max(a0, min(a, d[k])), min(a, d[k + 9]));
    }
    int b0 = -a0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const int j = k >> 1;
        const int b = max(max(max(pmax[j], pmax[j + 1]), pmax[j + 2]), pmax[j + 3]);
        b0 = min(min(b0, max(b, d[k])), max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// i / d for 0 <= i < 2^13 and 4 <= d <= 128 without an integer division per element (a runtime
// divisor costs ~30 VALU instructions; the loops below divide every pixel index): exact while
// i * (d - 1) < 2^20
struct SmallDiv {
    unsigned m;
    int d;
    __device__ explicit SmallDiv(int d_) : m(((1u << 20) + (unsigned)d_ - 1u) / (unsigned)d_), d(d_) {}
    __device__ int quot(int i) const { return (int)(((unsigned)i * m) >> 20); }
    __device__ void divmod(int i, int& q, int& r) const { q = quot(i); r = i - q * d; }
};

// Shapes of kf_detect_kernel: cells up to CW x CH, DET_LIST corners kept for the score pass (more: scored where they are
// found). The LDS of the window kernels decides how many of them share a CU with the alignment kernel: cells up to
// 56 x 48 (C2, C3) take 9 KB instead of the 18.4 KB of the largest shape.

template <int CW, int CH, int DET_LIST>
__global__ __launch_bounds__(256) void kf_detect_kernel(const DetectArgs* __restrict__ args) {
    constexpr int DET_TW = ((CW + 8 + 3 + 3) & ~3), DET_TH = CH + 8;   // cell + 4 px halo (+ up to 3 columns in front: rows staged from a dword boundary)
    constexpr int DET_RW = CW + 2, DET_RH = CH + 2;                  // raw scores: cell + 1 px
    const DetectArgs& a = args[blockIdx.z];
    if (a.enable && !*G(a.enable)) return;
    const int level = blockIdx.y;
    if (level >= a.n_levels) return;
    const ImgView im = a.level[level];
    const int gw = a.grid_w >> level, gh = a.grid_h >> level;
    if (gw <= 0 || gh <= 0 || gw > CW || gh > CH) return;
    const int ncx = im.w / gw, ncy = max(im.h / gh, 1);
    const int cell = blockIdx.x;
    if (cell >= ncx * ncy) return;
    const int cxi = cell % ncx, cyi = cell / ncx;
    const int left = cxi * gw, top = cyi * gh;
    const int tid = threadIdx.x;

    __shared__ __attribute__((aligned(16))) uint8_t s_t[DET_TH * DET_TW];
    __shared__ uint8_t s_raw[DET_RH * DET_RW];
    __shared__ unsigned s_key[4];
    __shared__ uint16_t s_list[DET_LIST];
    __shared__ int s_nlist;

    const int tw = gw + 8, th = gh + 8;
    const SmallDiv div_tw(tw), div_rw(gw + 2), div_gw(gw), div_gh(gh);
    // the cell + 4 px: inside the image whole dwords from the dword boundary left of its first column (column c of
    // the tile is byte ox + c of an LDS row), at the image border byte by byte with BORDER_REFLECT_101
    const int x0 = (left - 4) & ~3, ox = (left - 4) - x0;
    const bool inside = left - 4 >= 0 && top - 4 >= 0 && left + gw + 4 <= im.w && top + gh + 4 <= im.h && x0 + ((ox + tw + 3) & ~3) <= im.stride &&
                        (((reinterpret_cast<uintptr_t>(im.data) | (uintptr_t)im.stride) & 3) == 0);
    const uint8_t* const s_tb = s_t + (inside ? ox : 0);
    if (inside) {
        const int nq = (ox + tw + 3) >> 2;
        const SmallDiv div_nq(nq);
        SVO_GP(const uint8_t) g0 = im.g() + (size_t)(top - 4) * im.stride + x0;
        for (int i = tid; i < nq * th; i += 256) {
            int r, c;
            div_nq.divmod(i, r, c);
            *reinterpret_cast<uint32_t*>(&s_t[r * DET_TW + 4 * c]) = *reinterpret_cast<SVO_GP(const uint32_t)>(g0 + r * im.stride + 4 * c);
        }
    } else {
        for (int i = tid; i < tw * th; i += 256) {
            int r, c;
            div_tw.divmod(i, r, c);
            const int gy = reflect101(top - 4 + r, im.h), gx = reflect101(left - 4 + c, im.w);
            s_t[r * DET_TW + c] = im.g()[(size_t)gy * im.stride + gx];
        }
    }
    __syncthreads();
    // raw FAST scores on the cell + 1 px (0 outside the detector's 3 px border): corner test everywhere,
    // the corners gathered in LDS, their scores computed by all lanes together
    const int rw = gw + 2, rh = gh + 2;
    if (tid == 0) s_nlist = 0;
    __syncthreads();
    for (int i = tid; i < rw * rh; i += 256) {
        int r, c;
        div_rw.divmod(i, r, c);
        const int gy = top - 1 + r, gx = left - 1 + c;
        int sc = 0;
        if (gx >= 3 && gx < im.w - 3 && gy >= 3 && gy < im.h - 3 &&
            fast_is_corner(&s_tb[(r + 3) * DET_TW + c + 3], DET_TW, 6)) {
            const int slot = atomicAdd(&s_nlist, 1);
            if (slot < DET_LIST) s_list[slot] = (uint16_t)i;
            else sc = fast_corner_score(&s_tb[(r + 3) * DET_TW + c + 3], DET_TW, 6);
        }
        s_raw[r * DET_RW + c] = (uint8_t)sc;
    }
    __syncthreads();
    {
        const int nl = min(s_nlist, DET_LIST);
        for (int q = tid; q < nl; q += 256) {
            const int i = s_list[q];
            int r, c;
            div_rw.divmod(i, r, c);
            s_raw[r * DET_RW + c] = (uint8_t)fast_corner_score(&s_tb[(r + 3) * DET_TW + c + 3], DET_TW, 6);
        }
    }
    __syncthreads();
    // NMS + first-best in row-major order
    unsigned best = 0;
    for (int i = tid; i < gw * gh; i += 256) {
        int r, c;
        div_gw.divmod(i, r, c);
        if (top + r >= im.h) continue;
        const uint8_t* q = &s_raw[(r + 1) * DET_RW + c + 1];
        const int sc = q[0];
        if (sc && sc > q[1] && sc > q[-1] && sc > q[-DET_RW - 1] && sc > q[-DET_RW] &&
            sc > q[-DET_RW + 1] && sc > q[DET_RW - 1] && sc > q[DET_RW] && sc > q[DET_RW + 1]) {
            const unsigned key = ((unsigned)sc << 20) | (0xFFFFFu - (unsigned)i);
            best = max(best, key);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o, 64));
    if ((tid & 63) == 0) s_key[tid >> 6] = best;
    __syncthreads();
    best = max(max(s_key[0], s_key[1]), max(s_key[2], s_key[3]));
    int type = SVO_KP_FAST;
    float score, px, py;
    if (best) {
        const int i = (int)(0xFFFFFu - (best & 0xFFFFFu));
        score = (float)(best >> 20);
        px = (float)(left + i % gw);
        py = (float)(top + i / gw);
    } else {
        // edgelet: max of saturate_u8(Sobel dx), scan order x outer / y inner
        __syncthreads();
        unsigned eb = 0;
        for (int i = tid; i < gw * gh; i += 256) {
            int c, r;
            div_gh.divmod(i, c, r);                  // order index = x*gh + y
            const uint8_t* p1 = &s_tb[(r + 4) * DET_TW + c + 4];
            const uint8_t* p0 = p1 - DET_TW;
            const uint8_t* p2 = p1 + DET_TW;
            int v = (p0[1] - p0[-1]) + 2 * (p1[1] - p1[-1]) + (p2[1] - p2[-1]);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            const unsigned key = ((unsigned)v << 20) | (0xFFFFFu - (unsigned)i);
            eb = max(eb, key);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) eb = max(eb, (unsigned)__shfl_xor((int)eb, o, 64));
        if ((tid & 63) == 0) s_key[tid >> 6] = eb;
        __syncthreads();
        eb = max(max(s_key[0], s_key[1]), max(s_key[2], s_key[3]));
        const int i = (int)(0xFFFFFu - (eb & 0xFFFFFu));
        score = (float)(eb >> 20);
        px = (float)(left + i / gh);
        py = (float)(top + i % gh);
        type = SVO_KP_EDGELET;
    }
    if (tid == 0) {
        DetCell o;
        o.x = px; o.y = py; o.score = score; o.type = type;
        G(a.out)[(size_t)level * a.max_cells + cell] = o;
        if (cell == 0) G(a.n_out)[level] = ncx * ncy;
    }
}

void launch_detect(const DetectArgs* d_args, int batch, int max_cells, int n_levels, int grid_w, int grid_h, hipStream_t stream) {
    if (grid_w <= 56 && grid_h <= 48)
        hipLaunchKernelGGL((kf_detect_kernel<56, 48, 1024>), dim3(max_cells, n_levels, batch), dim3(256), 0, stream, d_args);
    else
        hipLaunchKernelGGL((kf_detect_kernel<DET_MAX_CW, DET_MAX_CH, 2048>), dim3(max_cells, n_levels, batch), dim3(256), 0, stream, d_args);
}

// --------------------------------------------------------- select + merge
__global__ __launch_bounds__(1024) void kf_select_merge_kernel(const MergeArgs* __restrict__ args) {
    const MergeArgs& a = args[blockIdx.x];
    if (a.enable && !*G(a.enable)) return;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nsel = G(a.n_det)[0];
    const int n_old = *G(a.kps.n);
    const int mcw = a.cam.grid_height, mch = a.cam.grid_width;   // swapped on purpose (:179-180)
    const int ncx = (a.width + mcw - 1) / mcw, ncy = (a.height + mch - 1) / mch;
    const int ncells = ncx * ncy;

    // select_best_keypoints: entry j of level i is compared with entry j of level 0
    for (int j = tid; j < nsel; j += nthr) {
        DetCell s = G(a.det)[j];
        int lvl = 0;
        for (int i = 1; i < a.n_levels; i++) {
            if (j >= G(a.n_det)[i]) continue;   // the reference would read out of bounds
            const DetCell c = G(a.det)[(size_t)i * a.max_cells + j];
            if (s.type == SVO_KP_FAST && c.type == SVO_KP_EDGELET) continue;
            if (s.type == c.type && s.score > c.score) continue;
            s = c;
            s.x *= (float)(1 << i);
            s.y *= (float)(1 << i);
            lvl = i;
        }
        G(a.sel)[j] = s;
        G(a.sel_level)[j] = lvl;
        G(a.sel_cell)[j] = -1;
    }
    for (int c = tid; c < ncells; c += nthr) G(a.occupied)[c] = 0;
    __syncthreads();
    // cells that already hold a keypoint (strictly inside)
    for (int i = tid; i < n_old; i += nthr) {
        const svo_kp2d k = G(a.kps.kps2d)[i];
        if (!(k.x > 0 && k.y > 0)) continue;
        const int xi = (int)floorf(k.x) / mcw, yi = (int)floorf(k.y) / mch;
        if (xi >= ncx || yi >= ncy) continue;
        const int l = xi * mcw, t = yi * mch;
        if (k.x > l && k.x < l + mcw && k.y > t && k.y < t + mch) G(a.occupied)[xi * ncy + yi] = 1;
    }
    __syncthreads();
    for (int j = tid; j < nsel; j += nthr) {
        const DetCell s = G(a.sel)[j];
        if (!(s.x > 0 && s.y > 0)) continue;
        const int xi = (int)floorf(s.x) / mcw, yi = (int)floorf(s.y) / mch;
        if (xi >= ncx || yi >= ncy) continue;
        const int l = xi * mcw, t = yi * mch;
        if (s.x > l && s.x < l + mcw && s.y > t && s.y < t + mch && !G(a.occupied)[xi * ncy + yi])
            G(a.sel_cell)[j] = xi * ncy + yi;
    }
    __syncthreads();
    // output order of the reference: cells x-outer / y-inner, candidates by index
    __shared__ int s_cnt;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int j = tid; j < nsel; j += nthr) {
        const int cj = G(a.sel_cell)[j];
        if (cj < 0) continue;
        int pos = 0;
        for (int q = 0; q < nsel; q++) {
            const int cq = G(a.sel_cell)[q];
            pos += (cq >= 0 && (cq < cj || (cq == cj && q < j))) ? 1 : 0;
        }
        atomicAdd(&s_cnt, 1);
        const int di = n_old + pos;
        if (di < a.cap) {
            const DetCell s = G(a.sel)[j];
            G(a.kps.kps2d)[di] = svo_kp2d{s.x, s.y};
            G(a.kps.score)[di] = s.score;
            G(a.kps.level_type)[di] = G(a.sel_level)[j] | (s.type << 8);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int n_new = s_cnt;
        if (n_old + n_new > a.cap) { n_new = a.cap - n_old; *G(a.overflow) = 1; }
        *G(a.old_count) = n_old;
        *G(a.kps.n) = n_old + n_new;
    }
}

void launch_select_merge(const MergeArgs* d_args, int batch, int max_cells, hipStream_t stream) {
    static const int env_threads = getenv("SVO_MERGE_THREADS") ? atoi(getenv("SVO_MERGE_THREADS")) : 0;   // (experiments)
    const int threads = env_threads >= 64 && env_threads <= 1024 && env_threads % 64 == 0 ? env_threads : (max_cells <= 512 ? 256 : 1024);
    hipLaunchKernelGGL(kf_select_merge_kernel, dim3(batch), dim3(threads), 0, stream, d_args);
}

// ------------------------------------------------------------------- init
__global__ __launch_bounds__(256) void kf_init_kernel(const KfInitArgs* __restrict__ args) {
    const KfInitArgs& a = args[blockIdx.x];
    if (a.enable && !*G(a.enable)) return;
    const int tid = threadIdx.x;
    const int n = *G(a.kps.n), old_count = *G(a.old_count);
    __shared__ PoseMats pm;
    __shared__ int s_cnt[4];
    float pose[6];
    for (int i = 0; i < 6; i++) pose[i] = a.first_frame ? 0.f : G(a.frame_pose)[i];
    if (tid == 0) {
        pose_mats(pose, pm);
        G(a.kfs)[a.new_kf_id] = a.record;                       // the keyframe table entry (pointers; n and pose below)
        if (a.evict_id >= 0) G(a.kfs)[a.evict_id].tmpl = nullptr;   // its template cache block now belongs to the new keyframe
    }
    if (a.record.tmpl_valid)
        for (int i = tid; i < a.tmpl_valid_bytes / 4; i += 256) ((SVO_GP(uint32_t))a.record.tmpl_valid)[i] = 0u;
    __syncthreads();
    const float fx = a.cam.fx, fy = a.cam.fy, cx = a.cam.cx, cy = a.cam.cy, baseline = a.cam.baseline;
    const uint32_t lcg0 = *G(a.color_lcg);
    for (int i = old_count + tid; i < n; i += 256) {
        const svo_kp2d kp = G(a.kps.kps2d)[i];
        const float disparity = G(a.disparity)[i];
        const float _z = baseline / fmaxf(0.5f, disparity);
        const float _x = (kp.x - cx) / fx * _z;
        const float _y = (kp.y - cy) / fy * _z;
        float loc[3] = {_x, _y, _z};
        mat33f_vec(pm.R, loc, loc);
        G(a.kps.kps3d)[i] = svo_kp3d{loc[0] + pm.t[0], loc[1] + pm.t[1], loc[2] + pm.t[2]};
        uint32_t st = lcg0;
        for (int q = old_count; q <= i; q++) st = st * 1664525u + 1013904223u;
        G(a.kps.color)[i] = (st >> 8) & 0xFFFFFFu;
        G(a.kps.kf_id)[i] = a.new_kf_id;
        G(a.kps.kp_index)[i] = i;
        G(a.kps.flags)[i] = SVO_IGNORE_TEMPORARY;
        G(a.kps.inl)[i] = 0;
        G(a.kps.outl)[i] = 0;
        const float deviation = (float)(0.5 / (double)(baseline / fx));
        G(a.kps.kfP)[i] = deviation * deviation;
        G(a.kps.kfx)[i] = 1 / _z;
    }
    __syncthreads();
    // keyframe.kps = frame.kps; keyframe.pose = frame.pose (keyframe_manager.cpp:27-29)
    const KfDev& kf = a.record;               // the pointers of the record (scalar loads)
    int not_temp = 0;
    for (int i = tid; i < n; i += 256) {
        const uint32_t f = G(a.kps.flags)[i];
        G(kf.kps2d)[i] = G(a.kps.kps2d)[i];
        G(kf.kps3d)[i] = G(a.kps.kps3d)[i];
        G(kf.flags)[i] = f;
        G(kf.outlier_count)[i] = G(a.kps.outl)[i];
        G(kf.inlier_count)[i] = G(a.kps.inl)[i];
        G(kf.kf_id)[i] = G(a.kps.kf_id)[i];
        G(kf.kp_index)[i] = G(a.kps.kp_index)[i];
        G(kf.score)[i] = G(a.kps.score)[i];
        G(kf.level_type)[i] = G(a.kps.level_type)[i];
        G(kf.color)[i] = G(a.kps.color)[i];
        G(kf.kfx)[i] = G(a.kps.kfx)[i];
        G(kf.kfP)[i] = G(a.kps.kfP)[i];
        not_temp += (f & SVO_IGNORE_TEMPORARY) ? 0 : 1;
    }
    not_temp = wave_sum_i(not_temp);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = not_temp;
    __syncthreads();
    not_temp = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (tid == 0) {
        SVO_GP(KfDev) kfw = G(a.kfs) + a.new_kf_id;
        kfw->n = n;
        for (int i = 0; i < 6; i++) kfw->pose[i] = pose[i];
        uint32_t st = lcg0;
        for (int q = old_count; q < n; q++) st = st * 1664525u + 1013904223u;
        *G(a.color_lcg) = st;
        *G(a.n_out) = n;
    }
    // stereo_slam.cpp:157-159 (first frame) and :237-245 (too few usable points)
    const bool clear_all = a.first_frame || ((size_t)not_temp < (size_t)n / 4);
    if (clear_all)
        for (int i = tid; i < n; i += 256) G(a.kps.flags)[i] &= ~(uint32_t)SVO_IGNORE_TEMPORARY;
}

void launch_kf_init(const KfInitArgs* d_args, int batch, hipStream_t stream) {
    hipLaunchKernelGGL(kf_init_kernel, dim3(batch), dim3(256), 0, stream, d_args);
}

}  // namespace svo
