// pyramid.hip — P1/P2 of SURVEY §8a: the two image pyramids of
// StereoSlam::new_image (src/lib/stereo_slam.cpp:131-140).
//
//  * pyr_halfsample_kernel: halfSample/createImgPyramid (stereo_slam.cpp:93-121),
//    all levels in ONE launch. A 64x64 level-0 tile is staged in LDS and
//    reduced level by level ((a+b+c+d)/4, truncating) — each level is written
//    once, level 0 is read once. HBM-bound streaming kernel.
//  * pyr_down_kernel: cv::pyrDown as used by cv::buildOpticalFlowPyramid
//    (stereo_slam.cpp:139): separable [1 4 6 4 1], BORDER_REFLECT_101,
//    (v+128)>>8, output ((w+1)/2, (h+1)/2). The Scharr derivative images of
//    OpenCV's pyramid are NOT materialised: klt.hip derives them from an LDS
//    tile, so the pyramid costs one byte written per pixel instead of five.
#include "svo_kernels.hpp"

namespace svo {

// ------------------------------------------------------------------ P1
constexpr int HS_TILE = 64;  // level-0 tile edge; yields levels up to 6 (1x1)

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

__global__ __launch_bounds__(256) void pyr_halfsample_kernel(const PyrArgs* __restrict__ args, int batch) {
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * HS_TILE, y0 = blockIdx.y * HS_TILE;
    if ((int)blockIdx.z >= batch) {
        // ingest of the right image: plain tile copy (createImgPyramid(right, 1), stereo_slam.cpp:136)
        const PyrArgs& b = args[blockIdx.z - batch];
        const int r = tid >> 2, c = (tid & 3) * 16;
        store_row16(b.dst_right, y0 + r, x0 + c, load_row16(b.src_right, y0 + r, x0 + c));
        return;
    }
    const PyrArgs& a = args[blockIdx.z];
    __shared__ uint8_t t0[HS_TILE * HS_TILE];
    __shared__ uint8_t t1[32 * 32];
    const bool ingest = a.src_left.data != nullptr;
    const ImgView src = ingest ? a.src_left : a.level[0];
    if (x0 >= src.w || y0 >= src.h) return;

    // stage the level-0 tile: 4 threads x 16 B per row, 64 rows
    {
        const int r = tid >> 2, c = (tid & 3) * 16;
        const uint4 v = load_row16(src, y0 + r, x0 + c);
        *reinterpret_cast<uint4*>(&t0[r * HS_TILE + c]) = v;
        if (ingest) store_row16(a.level[0], y0 + r, x0 + c, v);   // resident copy of level 0
    }
    __syncthreads();

    // level 1: 32x32 outputs, 4 per thread (one row segment of 4)
    const int n_levels = a.n_levels;
    if (n_levels > 1) {
        const ImgView d = a.level[1];
        const int r = tid >> 3, c = (tid & 7) * 4;
        uint8_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint8_t* p = &t0[(2 * r) * HS_TILE + 2 * (c + i)];
            o[i] = (uint8_t)((p[0] + p[1] + p[HS_TILE] + p[HS_TILE + 1]) / 4);
            t1[r * 32 + c + i] = o[i];
        }
        const int gy = (y0 >> 1) + r, gx = (x0 >> 1) + c;
        if (gy < d.h) {
            uint8_t* q = d.gw() + (size_t)gy * d.stride + gx;
            if (gx + 4 <= d.w && ((reinterpret_cast<uintptr_t>(q) & 3) == 0)) {
                *reinterpret_cast<uint32_t*>(q) = *reinterpret_cast<const uint32_t*>(o);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (gx + i < d.w) q[i] = o[i];
            }
        }
    }
    __syncthreads();

    // levels 2..: ping-pong between t0 (reused) and t1, one output per thread
    uint8_t* in = t1;
    uint8_t* out = t0;
    int edge = 32;  // edge of `in`
    for (int l = 2; l < n_levels && edge > 1; l++) {
        const int oe = edge >> 1;
        const ImgView d = a.level[l];
        if (tid < oe * oe) {
            const int r = tid / oe, c = tid % oe;
            const uint8_t* p = &in[(2 * r) * edge + 2 * c];
            const uint8_t v = (uint8_t)((p[0] + p[1] + p[edge] + p[edge + 1]) / 4);
            out[r * oe + c] = v;
            const int gy = (y0 >> l) + r, gx = (x0 >> l) + c;
            if (gy < d.h && gx < d.w) d.gw()[(size_t)gy * d.stride + gx] = v;
        }
        __syncthreads();
        uint8_t* tmp = in; in = out; out = tmp;
        edge = oe;
    }
}

// ------------------------------------------------------------------ P2
constexpr int PD_TW = 64, PD_TH = 16;                 // output tile
constexpr int PD_IH = 2 * PD_TH + 3;                  // input rows with halo
constexpr int PD_IWB = 2 * PD_TW + 8;                 // input bytes per row: columns 2*ox0-4 .. 2*ox0+131 (dword aligned)
constexpr int PD_IWD = PD_IWB / 4;                    // 34 dwords

// cv::pyrDown: separable [1 4 6 4 1] / 16 per axis, BORDER_REFLECT_101, (v + 128) >> 8.
// The input tile is staged with dword loads (byte loads + reflection only for the dwords that
// cross the left / right image border); the horizontal pass is one v_dot4 with (1,4,6,4) plus one byte per output,
// the vertical pass works on two 16-bit columns packed in a dword (row sums <= 4080, column
// sums <= 65280 < 2^16, so the halves never carry into each other).
__global__ __launch_bounds__(256) void pyr_down_kernel(const PyrArgs* __restrict__ args, int src_level) {
    const PyrArgs& a = args[blockIdx.z];
    const ImgView src = a.level[src_level];
    const ImgView dst = a.level[src_level + 1];
    __shared__ __attribute__((aligned(16))) uint32_t in[PD_IH][PD_IWD];
    __shared__ __attribute__((aligned(16))) uint32_t hrow[PD_IH][PD_TW / 2];    // 2 x u16 per dword
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * PD_TW, oy0 = blockIdx.y * PD_TH;
    if (ox0 >= dst.w || oy0 >= dst.h) return;
    const int ix0 = 2 * ox0 - 4, iy0 = 2 * oy0 - 2;

    const bool aligned = ((reinterpret_cast<uintptr_t>(src.data) | (uintptr_t)src.stride) & 3) == 0;
    for (int i = tid; i < PD_IH * PD_IWD; i += 256) {
        const int r = i / PD_IWD, d = i - r * PD_IWD;
        const uint8_t* row = src.g() + (size_t)reflect101(iy0 + r, src.h) * src.stride;
        const int x = ix0 + 4 * d;
        uint32_t v;
        if (aligned && x >= 0 && x + 4 <= src.w) {
            v = *reinterpret_cast<const uint32_t*>(row + x);
        } else {                                   // dwords on the left / right image border
            v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v |= (uint32_t)row[reflect101(x + b, src.w)] << (8 * b);
        }
        in[r][d] = v;
    }
    __syncthreads();
    // horizontal: 4 outputs c0..c0+3 per item from tile bytes 2*c0+2 .. 2*c0+12 (4 dwords)
    for (int i = tid; i < PD_IH * (PD_TW / 4); i += 256) {
        const int r = i / (PD_TW / 4), q = i - r * (PD_TW / 4);
        const uint32_t* w = &in[r][2 * q];
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        const uint32_t K = 0x04060401u;                                  // taps 1 4 6 4 (then + the 5th byte)
        const uint32_t h0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), K, (w1 >> 16) & 255u, false);
        const uint32_t h1 = __builtin_amdgcn_udot4(w1, K, w2 & 255u, false);
        const uint32_t h2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), K, (w2 >> 16) & 255u, false);
        const uint32_t h3 = __builtin_amdgcn_udot4(w2, K, w3 & 255u, false);
        hrow[r][2 * q] = h0 | (h1 << 16);
        hrow[r][2 * q + 1] = h2 | (h3 << 16);
    }
    __syncthreads();
    // vertical: 4 outputs per item (two packed dwords), stored as one dword
    for (int i = tid; i < PD_TH * (PD_TW / 4); i += 256) {
        const int r = i / (PD_TW / 4), q = i - r * (PD_TW / 4);
        uint32_t o[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = 2 * q + k;
            const uint32_t v = hrow[2 * r][c] + hrow[2 * r + 4][c] + 4u * (hrow[2 * r + 1][c] + hrow[2 * r + 3][c]) +
                               6u * hrow[2 * r + 2][c] + 0x00800080u;
            o[k] = ((v >> 8) & 255u) | ((v >> 24) << 8);              // two result bytes
        }
        const uint32_t out4 = o[0] | (o[1] << 16);
        const int gy = oy0 + r, gx = ox0 + 4 * q;
        if (gy < dst.h && gx < dst.w) {
            uint8_t* p = dst.gw() + (size_t)gy * dst.stride + gx;
            if (gx + 4 <= dst.w && ((reinterpret_cast<uintptr_t>(p) & 3) == 0)) {
                *reinterpret_cast<uint32_t*>(p) = out4;
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (gx + b < dst.w) p[b] = (uint8_t)(out4 >> (8 * b));
            }
        }
    }
}

void launch_pyr_halfsample(const PyrArgs* d_args, int batch, int w, int h, bool ingest, hipStream_t stream) {
    dim3 grid((w + HS_TILE - 1) / HS_TILE, (h + HS_TILE - 1) / HS_TILE, ingest ? 2 * batch : batch);
    hipLaunchKernelGGL(pyr_halfsample_kernel, grid, dim3(256), 0, stream, d_args, batch);
}

void launch_pyr_down(const PyrArgs* d_args, int batch, int src_level, int dst_w, int dst_h,
                     hipStream_t stream) {
    dim3 grid((dst_w + PD_TW - 1) / PD_TW, (dst_h + PD_TH - 1) / PD_TH, batch);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, stream, d_args, src_level);
}

}  // namespace svo
