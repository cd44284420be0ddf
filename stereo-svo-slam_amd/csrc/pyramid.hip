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

void launch_pyr_fused(const PyrArgs* d_args, int batch, int w, int h, bool right_blocks, hipStream_t stream) {
    dim3 grid((w + PF_T - 1) / PF_T, (h + PF_T - 1) / PF_T, right_blocks ? 2 * batch : batch);
    hipLaunchKernelGGL(pyr_fused_kernel, grid, dim3(256), 0, stream, d_args, batch);
}

}  // namespace svo
