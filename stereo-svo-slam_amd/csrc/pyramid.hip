// pyramid.hip — P1/P2 of SURVEY §8a: the two image pyramids of
// StereoSlam::new_image (src/lib/stereo_slam.cpp:131-140), built by ONE kernel that
// reads every level-0 pixel of the left image once:
//
//  * halfSample / createImgPyramid (stereo_slam.cpp:93-121): (a+b+c+d)/4, truncating,
//    all levels from a 64x64 level-0 tile in LDS;
//  * the image part of cv::buildOpticalFlowPyramid(left, .., Size(w,w), 2) (:139):
//    two levels of cv::pyrDown — separable [1 4 6 4 1], BORDER_REFLECT_101, (v+128)>>8,
//    output ((w+1)/2, (h+1)/2) — from the same tile, staged with a 6 pixel halo (level 1
//    needs 2 pixels of level 0 around its outputs, level 2 another 2 of level 1 = 4 of level
//    0). The Scharr derivative images of OpenCV's pyramid are NOT materialised: klt.hip
//    derives them from an LDS tile, so the pyramid costs one byte written per pixel, not five;
//  * optionally the ingest of the caller's device-resident frames (svo_new_images): the tile
//    is written to the resident level 0 while it is in LDS, the right image is copied by
//    extra workgroups.
//
// HBM-bound streaming kernel: per frame it moves 2WH bytes in (left + right), WH + WH out when
// it ingests, and the ~0.3 WH of pyramid levels. Integer arithmetic: bit exact.
#include "svo_kernels.hpp"
#include <cstdlib>

namespace svo {

constexpr int PF_T = 64;                 // level-0 tile edge; yields halfSample levels up to 6 (1x1)
constexpr int PF_HALO = 6;
constexpr int PF_ROWS = PF_T + 2 * PF_HALO;   // 76 tile rows: y0-6 .. y0+69
constexpr int PF_LS = 80;                // tile row stride: columns x0-8 .. x0+71 (dword aligned)
constexpr int PF_X0 = 8;                 // tile column of image column x0
constexpr int PF_L1 = 36;                // pyrDown level-1 block edge: indices x0/2-2 .. x0/2+33
constexpr int PF_L2 = 16;

// 16 bytes of row `gy` starting at column gx (0 beyond the row end)
__device__ inline uint4 load_row16(const ImgView& im, int gy, int gx) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gy < im.h && gx < im.w) {
        const uint8_t* p = im.g() + (size_t)gy * im.stride + gx;
        if (gx + 16 <= im.w && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
            v = *reinterpret_cast<const uint4*>(p);
        } else {
            uint8_t b[16];
#pragma unroll
            for (int i = 0; i < 16; i++) b[i] = (gx + i < im.w) ? p[i] : 0;
            v = *reinterpret_cast<const uint4*>(b);
        }
    }
    return v;
}
__device__ inline void store_row16(const ImgView& im, int gy, int gx, uint4 v) {
    if (gy >= im.h || gx >= im.w) return;
    uint8_t* q = im.gw() + (size_t)gy * im.stride + gx;
    if (gx + 16 <= im.w && ((reinterpret_cast<uintptr_t>(q) & 15) == 0)) {
        *reinterpret_cast<uint4*>(q) = v;
    } else {
        const uint8_t* b = reinterpret_cast<const uint8_t*>(&v);
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (gx + i < im.w) q[i] = b[i];
    }
}
// 4 output bytes of row gy at column gx (clipped at the row end)
__device__ inline void store_row4(const ImgView& im, int gy, int gx, uint32_t v) {
    if (gy >= im.h || gx >= im.w) return;
    uint8_t* q = im.gw() + (size_t)gy * im.stride + gx;
    if (gx + 4 <= im.w && ((reinterpret_cast<uintptr_t>(q) & 3) == 0)) {
        *reinterpret_cast<uint32_t*>(q) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (gx + i < im.w) q[i] = (uint8_t)(v >> (8 * i));
    }
}

__global__ __launch_bounds__(256) void pyr_fused_kernel(const PyrArgs* __restrict__ args, int batch) {
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * PF_T, y0 = blockIdx.y * PF_T;
    if ((int)blockIdx.z >= batch) {
        // ingest of the right image: plain tile copy (createImgPyramid(right, 1), stereo_slam.cpp:136)
        const PyrArgs& b = args[blockIdx.z - batch];
        if (!b.src_right.data) return;
        const int r = tid >> 2, c = (tid & 3) * 16;
        store_row16(b.dst_right, y0 + r, x0 + c, load_row16(b.src_right, y0 + r, x0 + c));
        return;
    }
    const PyrArgs& a = args[blockIdx.z];
    __shared__ __attribute__((aligned(16))) uint8_t t0[PF_ROWS * PF_LS];   // level-0 tile with halo
    __shared__ uint16_t hrow[PF_ROWS][PF_L1];                              // horizontal [1 4 6 4 1] sums
    __shared__ uint8_t lk1[PF_L1][PF_L1 + 4];                              // pyrDown level 1 of the tile (+ halo 2)
    __shared__ uint16_t hrow2[PF_L1][PF_L2];
    __shared__ uint8_t h1[32 * 32], h2[16 * 16];                           // halfSample ping-pong
    const bool ingest = a.src_left.data != nullptr;
    const ImgView src = ingest ? a.src_left : a.level[0];
    if (x0 >= src.w || y0 >= src.h) return;
    const int n_lk = a.n_lk;                         // LK levels to produce (<= 1: none)

    // ---- stage rows y0-6 .. y0+69, columns x0-8 .. x0+71 (BORDER_REFLECT_101 outside the image)
    {
        const bool aligned = ((reinterpret_cast<uintptr_t>(src.data) | (uintptr_t)src.stride) & 3) == 0;
        const int rows = n_lk > 1 ? PF_ROWS : PF_T;              // no halo rows without an LK pyramid
        const int r_lo = n_lk > 1 ? 0 : PF_HALO;
        // every thread's (up to) six dwords are requested before the first is stored: one round trip
        // to HBM per tile instead of six
        constexpr int NIT = (PF_ROWS * (PF_LS / 4) + 255) / 256;
        uint32_t v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; u++) {
            const int i = tid + 256 * u;
            v[u] = 0;
            if (i < rows * (PF_LS / 4)) {
                const int r = r_lo + i / (PF_LS / 4), d = i % (PF_LS / 4);
                const int gy = reflect101(y0 - PF_HALO + r, src.h);
                const uint8_t* row = src.g() + (size_t)gy * src.stride;
                const int x = x0 - PF_X0 + 4 * d;
                if (aligned && x >= 0 && x + 4 <= src.w) {
                    v[u] = *reinterpret_cast<const uint32_t*>(row + x);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++) v[u] |= (uint32_t)row[reflect101(x + b, src.w)] << (8 * b);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NIT; u++) {
            const int i = tid + 256 * u;
            if (i < rows * (PF_LS / 4)) {
                const int r = r_lo + i / (PF_LS / 4), d = i % (PF_LS / 4);
                *reinterpret_cast<uint32_t*>(&t0[r * PF_LS + 4 * d]) = v[u];
            }
        }
    }
    __syncthreads();
    const uint8_t* tc = &t0[PF_HALO * PF_LS + PF_X0];             // tile pixel (0,0) = image (x0, y0)

    // ---- resident copy of level 0 (ingest), 16 B per lane
    if (ingest && a.level[0].data != a.src_left.data) {
        const int r = tid >> 2, c = (tid & 3) * 16;
        if (y0 + r < src.h && x0 + c < src.w) {
            uint4 v;
            uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = *reinterpret_cast<const uint32_t*>(&tc[r * PF_LS + c + 4 * k]);
            store_row16(a.level[0], y0 + r, x0 + c, v);
        }
    }

    // ---- halfSample level 1: 32x32 outputs, 4 per thread
    const int n_levels = a.n_levels;
    if (n_levels > 1) {
        const ImgView d = a.level[1];
        const int r = tid >> 3, c = (tid & 7) * 4;
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint8_t* p = &tc[(2 * r) * PF_LS + 2 * (c + i)];
            const uint32_t v = (uint32_t)((p[0] + p[1] + p[PF_LS] + p[PF_LS + 1]) / 4);
            h1[r * 32 + c + i] = (uint8_t)v;
            o |= v << (8 * i);
        }
        store_row4(d, (y0 >> 1) + r, (x0 >> 1) + c, o);
    }
    // ---- pyrDown level 1, horizontal pass: tile rows 0..75, level-1 columns x0/2-2 .. x0/2+33
    if (n_lk > 1) {
        for (int i = tid; i < PF_ROWS * PF_L1; i += 256) {
            const int r = i / PF_L1, c = i - r * PF_L1;          // level-1 column x0/2 - 2 + c
            const uint8_t* p = &t0[r * PF_LS + 2 * c + 2];       // image column 2 * (x0/2 - 2 + c) - 2
            hrow[r][c] = (uint16_t)(p[0] + p[4] + 4 * (p[1] + p[3]) + 6 * p[2]);
        }
    }
    __syncthreads();

    // ---- pyrDown level 1, vertical pass -> lk1 block (+ its 32x32 centre to HBM)
    if (n_lk > 1) {
        const ImgView d = a.lk[1];
        for (int i = tid; i < PF_L1 * PF_L1; i += 256) {
            const int r = i / PF_L1, c = i - r * PF_L1;          // level-1 row y0/2 - 2 + r
            const int tr = 2 * r;                                // tile row of image row 2 * (y0/2 - 2 + r) - 2
            const uint32_t v = hrow[tr][c] + hrow[tr + 4][c] + 4u * (hrow[tr + 1][c] + hrow[tr + 3][c]) +
                               6u * hrow[tr + 2][c] + 128u;
            const uint8_t o = (uint8_t)(v >> 8);
            lk1[r][c] = o;
            const int gy = (y0 >> 1) + r - 2, gx = (x0 >> 1) + c - 2;
            if (r >= 2 && r < 34 && c >= 2 && c < 34 && gy < d.h && gx < d.w) d.gw()[(size_t)gy * d.stride + gx] = o;
        }
    }
    // ---- halfSample levels 2..: ping-pong between h1 and h2, one output per thread
    {
        uint8_t* in = h1;
        uint8_t* out = h2;
        int edge = 32;
        for (int l = 2; l < n_levels && edge > 1; l++) {
            const int oe = edge >> 1;
            const ImgView d = a.level[l];
            if (l > 2) __syncthreads();
            if (tid < oe * oe) {
                const int r = tid / oe, c = tid % oe;
                const uint8_t* p = &in[(2 * r) * edge + 2 * c];
                const uint8_t v = (uint8_t)((p[0] + p[1] + p[edge] + p[edge + 1]) / 4);
                out[r * oe + c] = v;
                const int gy = (y0 >> l) + r, gx = (x0 >> l) + c;
                if (gy < d.h && gx < d.w) d.gw()[(size_t)gy * d.stride + gx] = v;
            }
            uint8_t* tmp = in; in = out; out = tmp;
            edge = oe;
        }
    }
    if (n_lk <= 2) return;
    __syncthreads();

    // ---- pyrDown level 2 from the lk1 block (BORDER_REFLECT_101 on level-1 coordinates)
    {
        const ImgView d1 = a.lk[1], d2 = a.lk[2];
        const int bx = (x0 >> 1) - 2, by = (y0 >> 1) - 2;        // level-1 coordinates of lk1[0][0]
        for (int i = tid; i < PF_L1 * PF_L2; i += 256) {
            const int r = i / PF_L2, c = i - r * PF_L2;
            const int gy1 = by + r;                              // level-1 row of this block row
            uint32_t hs = 0;
            const int cx = 2 * ((x0 >> 2) + c);                  // level-1 column of the centre tap
            if (gy1 >= 0 && gy1 < d1.h && (x0 >> 2) + c < d2.w) {
                const int w[5] = {1, 4, 6, 4, 1};
#pragma unroll
                for (int t = 0; t < 5; t++) hs += w[t] * lk1[r][reflect101(cx - 2 + t, d1.w) - bx];
            }
            hrow2[r][c] = (uint16_t)hs;
        }
        __syncthreads();
        if (tid < PF_L2 * PF_L2) {
            const int r = tid / PF_L2, c = tid - r * PF_L2;
            const int gy = (y0 >> 2) + r, gx = (x0 >> 2) + c;
            if (gy < d2.h && gx < d2.w) {
                const int cy = 2 * gy;
                uint32_t v = 128u;
                const int w[5] = {1, 4, 6, 4, 1};
#pragma unroll
                for (int t = 0; t < 5; t++) v += w[t] * hrow2[reflect101(cy - 2 + t, d1.h) - by][c];
                d2.gw()[(size_t)gy * d2.stride + gx] = (uint8_t)(v >> 8);
            }
        }
    }
}

// ======================================================================================
// pyr_stream_kernel — the same two pyramids as pyr_fused_kernel, as a register-resident row
// stream (no LDS tile, no barrier): the fast path for frames whose width is a multiple of 8, whose
// height is even and whose rows are dword aligned (every configuration of the reference).
//
// A lane owns an 8-pixel column unit (two dwords per row); a wavefront owns 56 units (448 pixels)
// + one halo lane on either side, and walks down a block of RB level-0 rows (+ 6 above and 3 below
// for the vertical 5-tap windows of the two pyrDown levels). Per row a lane
//   * splits its dwords into even / odd pixels as packed u16 pairs (v_perm_b32),
//   * gets its neighbours' dwords through the wave (2 cross-lane moves),
//   * forms the horizontal [1 4 6 4 1] sums and the halfSample pair sums with packed 16-bit adds /
//     multiply-adds (every intermediate fits 16 bits: 16 * 255 and 256 * 255 + 128);
// every second row it emits one row of halfSample level 1 and of pyrDown level 1 (4 pixels = one
// dword store each), every fourth row level 2 of both (2 pixels), every eighth halfSample level 3
// (1 pixel); levels 4.. are reduced from the block's level-3 values through a 512-byte LDS tile at
// the end. BORDER_REFLECT_101: level-0 rows by address, level-0 columns inside the edge lanes,
// level-1 rows / columns (pyrDown level 2) by substitution inside the 5-tap window.
// HBM traffic = the level-0 image once (+ 14 % halo rows, which neighbouring blocks read at the same
// time) + the pyramid levels once: the kernel is bound by HBM, not by instruction issue
// (~6 VALU lane-operations per level-0 pixel against 45 in pyr_fused_kernel).
// ======================================================================================
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2u as_v2u(uint32_t v) { return __builtin_bit_cast(v2u, v); }
__device__ __forceinline__ uint32_t as_u32(v2u v) { return __builtin_bit_cast(uint32_t, v); }
// bytes: selector nibbles 0-3 = bytes of lo, 4-7 = bytes of hi, 0x0c = 0
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// the neighbouring lane's value through the wave-wide DPP shifts of gfx9 (one v_mov_b32_dpp, no LDS
// crossbar); lane 0 / lane 63 keep their own value
__device__ __forceinline__ uint32_t lane_up(uint32_t v) {      // value of lane - 1
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t lane_down(uint32_t v) {    // value of lane + 1
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

constexpr int PS_UNITS = 56;     // useful 8-pixel units per wavefront (lanes 1..56; a multiple of 8: level 6 = 64 pixels)

// (x0, x2) + 4 (x1, x3) ... : [1 4 6 4 1] over five packed pairs
__device__ __forceinline__ v2u tap5(v2u a, v2u b, v2u c, v2u d, v2u e) {
    const v2u four = {4, 4}, six = {6, 6};
    return (a + e) + four * (b + d) + six * c;
}
// the 5-tap window slot (offset -2..2 around the centre) that BORDER_REFLECT_101 maps tap `t` to
__device__ __forceinline__ v2u pick5(int off, v2u m2, v2u m1, v2u z, v2u p1, v2u p2) {
    return off == -2 ? m2 : off == -1 ? m1 : off == 0 ? z : off == 1 ? p1 : p2;
}

// RB: level-0 rows per workgroup. 64 = one row of level 6 (seven levels). Up to six levels 32 will do (one
// row of level 5): twice the wavefronts — 8 per SIMD instead of 4 at C2 with 256 frames, every one with its
// six rows in flight — for 10 more halo rows per 64, which the neighbour block reads at the same time (L2).
template <int RB>
__global__ __launch_bounds__(64) void pyr_stream_kernel(const PyrArgs* __restrict__ args, int batch) {
    static_assert(RB == 32 || RB == 64, "row block");
    const int lane = threadIdx.x;
    const int y0 = blockIdx.y * RB;
    if ((int)blockIdx.z >= batch) {
        // ingest of the right image: plain copy of the block's rows (createImgPyramid(right, 1), stereo_slam.cpp:136)
        const PyrArgs& b = args[blockIdx.z - batch];
        if (!b.src_right.data) return;
        const int x = (blockIdx.x * 64 + lane) * 8;
        if (x >= b.src_right.w) return;
        const int y1 = min(y0 + RB, b.src_right.h);
        for (int y = y0; y < y1; y++) {
            const uint2 v = *reinterpret_cast<SVO_GP(const uint2)>(b.src_right.g() + (size_t)y * b.src_right.stride + x);
            *reinterpret_cast<SVO_GP(uint2)>(b.dst_right.gw() + (size_t)y * b.dst_right.stride + x) = v;
        }
        return;
    }
    const PyrArgs& a = args[blockIdx.z];
    const bool ingest = a.src_left.data != nullptr;
    const ImgView src = ingest ? a.src_left : a.level[0];
    const int W = src.w, H = src.h;
    const int U = W >> 3;                                  // 8-pixel units per row
    if (y0 >= H || (int)blockIdx.x * PS_UNITS >= U) return;
    const int u = (int)blockIdx.x * PS_UNITS - 1 + lane;   // this lane's unit (lanes 0 and 57: halo)
    const bool useful = lane >= 1 && lane <= PS_UNITS && u < U;
    const int uc = min(max(u, 0), U - 1);
    const bool first_unit = u == 0, last_unit = u == U - 1;
    const int n_levels = a.n_levels, n_lk = a.n_lk;
    const bool copy0 = ingest && a.level[0].data != a.src_left.data;
    const ImgView lv0 = a.level[0], hs1v = a.level[1], hs2v = a.level[2], hs3v = a.level[3], lk1v = a.lk[1], lk2v = a.lk[2];
    const int h1 = (H + 1) >> 1;                           // pyrDown level-1 rows
    const int h2 = (h1 + 1) >> 1;
    const int y_end = min(y0 + RB, H);

    __shared__ uint8_t s_l3[RB / 8][64];                    // halfSample level 3 of the block, column = lane
    __shared__ uint8_t s_l4[RB / 16][32];
    __shared__ uint8_t s_l5[RB / 32][16];

    SVO_GP(const uint8_t) col = src.g() + 8 * uc;
    auto load_row = [&](int y) -> uint2 {
        const int gy = reflect101(y, H);
        return *reinterpret_cast<SVO_GP(const uint2)>(col + (size_t)gy * src.stride);
    };
    auto copy_row = [&](int y, uint2 d) {                   // resident copy of level 0 (ingest)
        if (copy0 && useful && y >= y0 && y < y_end)
            *reinterpret_cast<SVO_GP(uint2)>(lv0.gw() + (size_t)y * lv0.stride + 8 * u) = d;
    };
    // horizontal pass of one level-0 row: [1 4 6 4 1] sums centred on the even pixels (two pairs) and
    // the horizontal pair sums of halfSample
    struct RowOut { v2u h01, h23, s01, s23; };
    auto row_pass = [&](uint2 d) -> RowOut {
        uint32_t L = lane_up(d.y), R = lane_down(d.x);
        if (first_unit) L = perm(0, d.x, 0x01020c0cu);        // p[-2] = p[2], p[-1] = p[1]
        if (last_unit) R = perm(0, d.y, 0x0c0c0c02u);         // p[8] = p[6]
        const v2u e0 = as_v2u(perm(0, d.x, 0x0c020c00u)), o0 = as_v2u(perm(0, d.x, 0x0c030c01u));
        const v2u e1 = as_v2u(perm(0, d.y, 0x0c020c00u)), o1 = as_v2u(perm(0, d.y, 0x0c030c01u));
        const v2u am = as_v2u(perm(d.x, L, 0x0c040c02u));     // (p[-2], p0)
        const v2u bm = as_v2u(perm(d.x, L, 0x0c050c03u));     // (p[-1], p1)
        const v2u ep = as_v2u(perm(d.y, d.x, 0x0c040c02u));   // (p2, p4)
        const v2u b2 = as_v2u(perm(d.y, d.x, 0x0c050c03u));   // (p3, p5)
        const v2u e2 = as_v2u(perm(R, d.y, 0x0c040c02u));     // (p6, p8)
        RowOut o;
        o.h01 = tap5(am, bm, e0, o0, ep);
        o.h23 = tap5(ep, b2, e1, o1, e2);
        o.s01 = e0 + o0;
        o.s23 = e1 + o1;
        return o;
    };

    // Level-1 rows q of this block: rows [y0/2, (y0+RB)/2) of halfSample / pyrDown level 1 are stored;
    // pyrDown level 2 needs level-1 rows from y0/2 - 2 up to (y0+RB)/2 (its vertical window). Step q takes
    // level-0 rows 2q+1 and 2q+2 in: the window then holds the horizontal sums of rows 2q-2 .. 2q+2.
    const int q_store_lo = y0 >> 1, q_store_hi = min((y0 + RB) >> 1, h1);
    const int q_lo = n_lk > 2 ? q_store_lo - 2 : q_store_lo;
    const int q_hi = n_lk > 2 ? min((y0 + RB) >> 1, h1 + 1) : q_store_hi - 1;     // inclusive
    v2u ha0, ha1, ha2, ha3, ha4, hb0, hb1, hb2, hb3, hb4;
    v2u s01_prev, s23_prev;                                  // pair sums of level-0 row 2q (halfSample)
    {
        const uint2 d0 = load_row(2 * q_lo - 2), d1 = load_row(2 * q_lo - 1), d2 = load_row(2 * q_lo);
        copy_row(2 * q_lo - 2, d0); copy_row(2 * q_lo - 1, d1); copy_row(2 * q_lo, d2);
        const RowOut r0 = row_pass(d0), r1 = row_pass(d1), r2 = row_pass(d2);
        ha2 = r0.h01; hb2 = r0.h23; ha3 = r1.h01; hb3 = r1.h23; ha4 = r2.h01; hb4 = r2.h23;
        ha0 = ha1 = hb0 = hb1 = ha2;
        s01_prev = r2.s01; s23_prev = r2.s23;
    }
    v2u t01_prev = {0, 0}, t23_prev = {0, 0};               // halfSample level-1 row q - 1 as pairs
    int hs2_prev0 = 0, hs2_prev1 = 0;                       // halfSample level-2 row (q >> 1) - 1
    v2u wm4 = {0, 0}, wm3 = {0, 0}, wm2 = {0, 0}, wm1 = {0, 0}, w0 = {0, 0};   // level-1 rows q-4 .. q, horizontal sums of level 2
    // level-0 rows are requested three steps (six rows) before they are used
    uint2 nx0 = load_row(2 * q_lo + 1), nx1 = load_row(2 * q_lo + 2), nx2 = load_row(2 * q_lo + 3),
          nx3 = load_row(2 * q_lo + 4), nx4 = load_row(2 * q_lo + 5), nx5 = load_row(2 * q_lo + 6);
    const v2u r128 = {128, 128}, eight = {8, 8}, two = {2, 2};

    for (int q = q_lo; q <= q_hi; q++) {
        const uint2 d1 = nx0, d2 = nx1;                     // level-0 rows 2q+1, 2q+2
        nx0 = nx2; nx1 = nx3; nx2 = nx4; nx3 = nx5;
        nx4 = load_row(2 * q + 7); nx5 = load_row(2 * q + 8);
        copy_row(2 * q + 1, d1); copy_row(2 * q + 2, d2);
        const RowOut ra = row_pass(d1), rb = row_pass(d2);
        // ---- halfSample level 1, row q: level-0 rows 2q, 2q+1; levels 2 and 3 every second / fourth row
        if (q >= q_store_lo && q < q_store_hi && 2 * q + 1 < H && n_levels > 1) {
            const v2u t01 = (s01_prev + ra.s01) >> two, t23 = (s23_prev + ra.s23) >> two;
            if (useful && q < hs1v.h)
                *reinterpret_cast<SVO_GP(uint32_t)>(hs1v.gw() + (size_t)q * hs1v.stride + 4 * u) = perm(as_u32(t23), as_u32(t01), 0x06040200u);
            if ((q & 1) && n_levels > 2) {          // halfSample level 2, row q >> 1: level-1 rows q-1, q
                const uint32_t ua = as_u32(t01_prev + t01), ub = as_u32(t23_prev + t23);
                const int v0 = (int)((ua & 0xffffu) + (ua >> 16)) >> 2;
                const int v1 = (int)((ub & 0xffffu) + (ub >> 16)) >> 2;
                const int r2 = q >> 1;
                if (useful && r2 < hs2v.h)
                    *reinterpret_cast<SVO_GP(uint16_t)>(hs2v.gw() + (size_t)r2 * hs2v.stride + 2 * u) = (uint16_t)(v0 | (v1 << 8));
                if ((r2 & 1) && n_levels > 3) {     // halfSample level 3, row q >> 2
                    const int v3 = (hs2_prev0 + hs2_prev1 + v0 + v1) >> 2;
                    const int r3 = q >> 2;
                    if (useful && r3 < hs3v.h && u < hs3v.w) hs3v.gw()[(size_t)r3 * hs3v.stride + u] = (uint8_t)v3;
                    s_l3[r3 - (y0 >> 3)][lane] = (uint8_t)v3;
                }
                hs2_prev0 = v0; hs2_prev1 = v1;
            }
            t01_prev = t01; t23_prev = t23;
        }
        s01_prev = rb.s01; s23_prev = rb.s23;
        // ---- pyrDown level 1, row q: the window moves on to level-0 rows 2q-2 .. 2q+2
        ha0 = ha2; ha1 = ha3; ha2 = ha4; ha3 = ra.h01; ha4 = rb.h01;
        hb0 = hb2; hb1 = hb3; hb2 = hb4; hb3 = ra.h23; hb4 = rb.h23;
        v2u nw = {0, 0};
        if (n_lk > 1 && q >= 0 && q < h1) {
            const v2u c01 = (tap5(ha0, ha1, ha2, ha3, ha4) + r128) >> eight;
            const v2u c23 = (tap5(hb0, hb1, hb2, hb3, hb4) + r128) >> eight;
            const uint32_t P = perm(as_u32(c23), as_u32(c01), 0x06040200u);     // level-1 pixels 4u .. 4u+3
            if (useful && q >= q_store_lo && q < q_store_hi)
                *reinterpret_cast<SVO_GP(uint32_t)>(lk1v.gw() + (size_t)q * lk1v.stride + 4 * u) = P;
            if (n_lk > 2) {
                // horizontal pass of pyrDown level 2 on this level-1 row (BORDER_REFLECT_101 on level-1 columns)
                uint32_t PL = lane_up(P), PR = lane_down(P);
                if (first_unit) PL = perm(0, P, 0x01020c0cu);
                if (last_unit) PR = perm(0, P, 0x0c0c0c02u);
                nw = tap5(as_v2u(perm(P, PL, 0x0c040c02u)), as_v2u(perm(P, PL, 0x0c050c03u)),
                          as_v2u(perm(0, P, 0x0c020c00u)), as_v2u(perm(0, P, 0x0c030c01u)),
                          as_v2u(perm(PR, P, 0x0c040c02u)));
            }
        }
        if (n_lk > 2) {
            wm4 = wm3; wm3 = wm2; wm2 = wm1; wm1 = w0; w0 = nw;    // level-1 rows q-4 .. q
            // pyrDown level 2, row c / 2 with the centre level-1 row c = q - 2 (taps q-4 .. q)
            const int c = q - 2;
            if (c >= 0 && !(c & 1) && (c >> 1) >= (y0 >> 2) && (c >> 1) < min((y0 + RB) >> 2, h2)) {
                v2u t0 = wm4, t1 = wm3, t3 = wm1, t4 = w0;
                if (c - 2 < 0 || c + 2 >= h1) {         // BORDER_REFLECT_101 on level-1 rows: the tap is a row of the window
                    t0 = pick5(reflect101(c - 2, h1) - c, wm4, wm3, wm2, wm1, w0);
                    t1 = pick5(reflect101(c - 1, h1) - c, wm4, wm3, wm2, wm1, w0);
                    t3 = pick5(reflect101(c + 1, h1) - c, wm4, wm3, wm2, wm1, w0);
                    t4 = pick5(reflect101(c + 2, h1) - c, wm4, wm3, wm2, wm1, w0);
                }
                const uint32_t o = as_u32((tap5(t0, t1, wm2, t3, t4) + r128) >> eight);
                if (useful && 2 * u < lk2v.w)
                    *reinterpret_cast<SVO_GP(uint16_t)>(lk2v.gw() + (size_t)(c >> 1) * lk2v.stride + 2 * u) =
                        (uint16_t)((o & 0xffu) | ((o >> 8) & 0xff00u));
            }
        }
    }
    // ---- halfSample levels 4.. from the block's level-3 values (one wavefront: LDS in program order)
    if (n_levels > 4) {
        __builtin_amdgcn_wave_barrier();
        const int u0 = (int)blockIdx.x * PS_UNITS;
        const int rows3 = max(min((y0 + RB) >> 3, hs3v.h) - (y0 >> 3), 0);
        {
            const ImgView d = a.level[4];
            const int c = lane & 31;                            // 28 columns x RB/16 rows
            for (int rr = lane >> 5; rr < RB / 16; rr += 2) {
                if (c < PS_UNITS / 2 && 2 * rr + 1 < rows3) {
                    const int v = (s_l3[2 * rr][1 + 2 * c] + s_l3[2 * rr][2 + 2 * c] + s_l3[2 * rr + 1][1 + 2 * c] + s_l3[2 * rr + 1][2 + 2 * c]) >> 2;
                    s_l4[rr][c] = (uint8_t)v;
                    const int gx = (u0 >> 1) + c, gy = (y0 >> 4) + rr;
                    if (gx < d.w && gy < d.h) d.gw()[(size_t)gy * d.stride + gx] = (uint8_t)v;
                }
            }
        }
        if (n_levels > 5) {
            __builtin_amdgcn_wave_barrier();
            const ImgView d = a.level[5];
            const int c = lane & 15, r = lane >> 4;             // 14 columns x RB/32 rows
            const int gx = (u0 >> 2) + c, gy = (y0 >> 5) + r;
            if (c < PS_UNITS / 4 && r < RB / 32 && gx < d.w && gy < d.h) {
                const int v = (s_l4[2 * r][2 * c] + s_l4[2 * r][2 * c + 1] + s_l4[2 * r + 1][2 * c] + s_l4[2 * r + 1][2 * c + 1]) >> 2;
                s_l5[r][c] = (uint8_t)v;
                d.gw()[(size_t)gy * d.stride + gx] = (uint8_t)v;
            }
            if (RB >= 64 && n_levels > 6) {
                __builtin_amdgcn_wave_barrier();
                const ImgView d6 = a.level[6];
                const int gx6 = (u0 >> 3) + lane, gy6 = y0 >> 6;
                if (lane < PS_UNITS / 8 && gx6 < d6.w && gy6 < d6.h) {
                    const int v = (s_l5[0][2 * lane] + s_l5[0][2 * lane + 1] + s_l5[RB / 64][2 * lane] + s_l5[RB / 64][2 * lane + 1]) >> 2;
                    d6.gw()[(size_t)gy6 * d6.stride + gx6] = (uint8_t)v;
                }
            }
        }
    }
}

// 0, or the row block of the streaming kernel when its conditions hold: 8-pixel units, even height, aligned rows everywhere it
// uses wide accesses (every buffer the tracker allocates; a caller's image when its pointer and
// stride are multiples of 8)
int pyr_stream_rows(const PyrArgs& a) {
    auto al = [](const ImgView& v, uintptr_t m) { return ((reinterpret_cast<uintptr_t>(v.data) | (uintptr_t)v.stride) & m) == 0; };
    const bool ingest = a.src_left.data != nullptr;
    const ImgView& src = ingest ? a.src_left : a.level[0];
    if (src.w % 8 || src.h % 2 || src.w < 16 || src.h < 16 || !al(src, 7)) return 0;
    if (a.n_levels > 7 || a.n_lk > 3) return 0;
    if (ingest && a.level[0].data != a.src_left.data && !al(a.level[0], 7)) return 0;
    if (a.src_right.data && (!al(a.src_right, 7) || !al(a.dst_right, 7))) return 0;
    if (a.n_levels > 1 && !al(a.level[1], 3)) return 0;
    if (a.n_levels > 2 && !al(a.level[2], 1)) return 0;
    if (a.n_lk > 1 && !al(a.lk[1], 3)) return 0;
    if (a.n_lk > 2 && !al(a.lk[2], 1)) return 0;
    return a.n_levels > 6 ? 64 : 32;
}

void launch_pyr_fused(const PyrArgs* d_args, int batch, int w, int h, bool right_blocks, int stream_rows, hipStream_t stream) {
    const char* env = getenv("SVO_PYR_KERNEL");          // "tile": always the tile kernel; "64": 64-row blocks (tests, A/B runs)
    if (stream_rows > 0 && !(env && env[0] == 't')) {
        const int units = w / 8;
        if (env && env[0] == '6') stream_rows = 64;
        dim3 grid((units + PS_UNITS - 1) / PS_UNITS, (h + stream_rows - 1) / stream_rows, right_blocks ? 2 * batch : batch);
        if (stream_rows == 32) hipLaunchKernelGGL(pyr_stream_kernel<32>, grid, dim3(64), 0, stream, d_args, batch);
        else hipLaunchKernelGGL(pyr_stream_kernel<64>, grid, dim3(64), 0, stream, d_args, batch);
        return;
    }
    dim3 grid((w + PF_T - 1) / PF_T, (h + PF_T - 1) / PF_T, right_blocks ? 2 * batch : batch);
    hipLaunchKernelGGL(pyr_fused_kernel, grid, dim3(256), 0, stream, d_args, batch);
}

}  // namespace svo
