// klt.hip — row B2 of SURVEY §8a: OpticalFlow::calculate_optical_flow
// (src/lib/optical_flow.cpp:14-56) = cv::calcOpticalFlowPyrLK with a (w,w)
// window, maxLevel 2, 30 iterations / eps 0.01, OPTFLOW_USE_INITIAL_FLOW,
// tracking every keypoint from the Gaussian pyramid of its ORIGIN KEYFRAME
// into the current frame (PoseRefiner::refine_pose, src/lib/pose_refinement.cpp:72-118).
//
// One wavefront per keypoint, all pyramid levels and iterations inside the
// kernel. A thread owns one window column and RPT consecutive rows, so
//  * the bilinear taps slide down the rows, and every 4-tap interpolation is
//    two v_dot2_i32_i16 (14-bit weights and pixels / derivatives all fit int16);
//  * the fixed-point template (I, Ix, Iy; 5 fractional bits) of the thread's
//    rows lives in REGISTERS as int16 row pairs for all iterations of a level;
//    b = sum (J - I) dI is evaluated as sum J dI minus the level's constant sum I dI.
// The template depends on the keyframe only: a tracker builds it once per
// keypoint and level (reference neighbourhood staged in LDS, Scharr derivatives
// of OpenCV's pyramid computed from the tile: zero outside the image, the
// constant border of cv::buildOpticalFlowPyramid) and keeps it in HBM
// (KfDev::tmpl); later frames load it. The search tile of the current frame is
// staged per level as 16-bit values in two copies one pixel apart, so that a
// bilinear pair is one aligned ds_read_b32. An iteration touches HBM only when
// the window drifts out of the search tile. All window sums are exact integers,
// so the result does not depend on the reduction order.
#include "svo_kernels.hpp"
#include <cstring>
#include <utility>

namespace svo {

constexpr int KLT_MAX_WIN = 35;                        // (the entry points reject larger windows)
static_assert(KLT_MAX_WIN + 1 <= 36, "largest kernel shape");
// Pixels the window may drift before the search tile is staged again. The result does not depend on it (exact
// integer sums from whatever tile holds the window); 4 instead of 6 makes the 32-column shape's tile 8.1 instead of
// 9.3 KB — LDS is what the window kernels of a step compete for — and a level still stages once (+0.9 % frames/s).
#ifndef SVO_KLT_MARGIN
#define SVO_KLT_MARGIN 4
#endif
constexpr int KLT_MARGIN = SVO_KLT_MARGIN;
// The search tile is kept as 16-bit values (pixel << 7) in TWO copies, the second shifted by one pixel:
// the bilinear pair (p[x], p[x+1]) of any column x is then ONE aligned ds_read_b32 (copy x & 1, dword
// x >> 1) that already is the int16 pair v_dot2 wants — no byte loads, no packing in the iteration.
// (Unaligned ds_read_b32 stall the LDS pipeline: SQ_LDS_UNALIGNED_STALL.) The << 7 turns the
// (sum + 2^8) >> 9 of the fixed-point interpolation into a >> 16: the two results of a row pair are
// the high halves of two dwords, one v_perm_b32. ds_read_b32 serves lanes 0-31 and 32-63 in one LDS
// cycle each, banks = dword address mod 32: a row group's 32 columns read 16 consecutive dwords of either
// copy, so the second copy lies 16 banks (mod 32) from the first. (Until round 3 it was 32 on: every window
// read a two-way conflict.)
// Sizes per kernel shape (CW = 32: windows up to 31, CW = 36: up to 35). The LDS of a workgroup is what
// bounds the wavefronts per CU of this kernel (125 registers: 16), so the 32-column shape does not pay
// for the 36-column one: 9.3 KB instead of 12.8 (12 -> 16 wavefronts per CU).
template <int CW>
struct KltGeom {
    static constexpr int DW = CW;                            // derivative / tap grid edge (w+1 <= CW)
    static constexpr int RS = 44;                            // row stride of the reference tile: (w+3) + 3 alignment slack, /4
    static constexpr int RROWS = CW + 2;                     // w+3 rows; also >= 1 + rows covered by the row groups + 1
    static constexpr int TJROWS = DW + 2 * KLT_MARGIN + 2;   // + slack: idle rows of the last row group are read, not used
    static constexpr int TJS = (DW + 2 * KLT_MARGIN + 6) & ~3;   // columns of the search tile: (w+1) + 2*margin + 3, /4
    static constexpr int J2S = (TJS / 2) | 1;                // dwords per tile row of one copy
    static constexpr int J2COPY = ((TJROWS * J2S + 31) / 32) * 32 + 16;   // dwords between the two copies
    static constexpr int SD = (DW + 1) * DW;                 // ints of the derivative grid (+ one slack row)
    static constexpr int SJ2 = J2COPY + TJROWS * J2S;        // dwords of the two-copy search tile
    static constexpr int SI = (RROWS * RS + 3) / 4;          // dwords of the reference tile
    static constexpr int LDS_DWORDS = SD + SI > SJ2 ? SD + SI : SJ2;
    static_assert(TJS <= 2 * J2S, "search tile row");
};
#ifndef SVO_KLT_THREADS
#define SVO_KLT_THREADS 64          // 128: two waves per keypoint (4 row groups of 8 rows)
#endif
constexpr int KLT_THREADS = SVO_KLT_THREADS;
constexpr int KLT_WAVES = KLT_THREADS / 64;

typedef short v2s __attribute__((ext_vector_type(2)));

// Diagnostic builds (-DSVO_KLT_PHASES, tools/klt_phases.py): cycles of thread 0 per phase of the kernel,
// summed over all wavefronts since the last read
#ifdef SVO_KLT_PHASES
// (1024 sets of counters, picked by the workgroup index: one set for all wavefronts made the atomics the
// slowest part of the kernel)
__device__ unsigned long long g_klt_phases[1024][16];
#define KLT_SLOT_ ((blockIdx.x + 131u * blockIdx.y) & 1023u)
#define KLT_PHASE(i)                                                                        \
    do {                                                                                    \
        const unsigned long long t_now_ = __builtin_amdgcn_s_memtime();                     \
        if (threadIdx.x == 0) atomicAdd(&g_klt_phases[KLT_SLOT_][i], t_now_ - t_phase_);    \
        t_phase_ = t_now_;                                                                  \
    } while (0)
#define KLT_COUNT(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_klt_phases[KLT_SLOT_][i], (unsigned long long)(v)); } while (0)
#define KLT_PHASE_WAIT(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); KLT_PHASE(i); } while (0)
#else
#define KLT_PHASE_WAIT(i) do { } while (0)
#define KLT_PHASE(i) do { } while (0)
#define KLT_COUNT(i, v) do { } while (0)
#endif


__device__ inline v2s as_v2s(int v) { return __builtin_bit_cast(v2s, v); }
__device__ inline int as_int(v2s v) { return __builtin_bit_cast(int, v); }
__device__ inline int pack16(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
// a.x*b.x + a.y*b.y + c, exact int32
__device__ inline int dot2(v2s a, v2s b, int c) { return __builtin_amdgcn_sdot2(a, b, c, false); }
// the same with a wave-uniform c taken from a scalar register: the compiler only selects the accumulate-in-place
// form (v_dot2c), which costs a v_mov of the constant into the destination first — 16 per window pass
__device__ inline int dot2_uc(v2s a, v2s b, int c_uniform) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
}

// LDS byte pairs. The window columns start at an arbitrary byte of the tile, and unaligned
// ds_read_u16/b32 stall the LDS pipeline (SQ_LDS_UNALIGNED_STALL ~ its whole active time when
// measured), while the compiler fuses adjacent byte loads into exactly those. So the tiles are
// read with explicit ds_read_u8 (d16_hi loads do not keep the other half with SRAM-ECC on, so
// a pair costs two loads and one v_lshl_or). The loads are issued without waiting;
// lds_fence() must run before the registers are used.
__device__ inline uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)((const SVO_LDS(uint8_t)*)p);
}
template <int OFF>
__device__ inline int lds_byte_nowait(uint32_t addr) {
    int r;
    asm volatile("ds_read_u8 %0, %1 offset:%2" : "=&v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int N>
__device__ inline void lds_fence(int (&r)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(r[i]));
}
template <int STRIDE, int OFF, int... U>
__device__ inline void lds_col_impl(uint32_t a, int* r, std::integer_sequence<int, U...>) {
    ((r[U] = lds_byte_nowait<U * STRIDE + OFF>(a)), ...);
}
template <int STRIDE, int OFF, int N>
__device__ inline void lds_col_nowait(uint32_t a, int (&r)[N]) {
    lds_col_impl<STRIDE, OFF>(a, r, std::make_integer_sequence<int, N>{});
}
// r[u] = (tile[u*STRIDE + LO], tile[u*STRIDE + HI]) as int16 pairs, u = 0..N-1 (waits for the loads)
template <int STRIDE, int LO, int HI, int N>
__device__ inline void lds_pairs(uint32_t a, int (&r)[N]) {
    int hi[N];
    lds_col_nowait<STRIDE, LO>(a, r);
    lds_col_nowait<STRIDE, HI>(a, hi);
    lds_fence(r);
    lds_fence(hi);
#pragma unroll
    for (int u = 0; u < N; u++) r[u] |= hi[u] << 16;
}

// exact sums of NV per-thread int32 partials over the wavefront as doubles (integers below 2^53:
// exact; (float) of one rounds once, like (float) of the 64-bit integer sum). Each partial is split
// into its low 16 bits and the signed high part (the 64-lane sum of either fits int32) and the 2 NV
// reductions run step by step side by side: a DPP add needs two wait states behind the write of its
// source, which the other chains fill (one chain after the other left an s_nop behind every add).
template <int CTRL, int N>
__device__ inline void dpp_add_all(int (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_i<CTRL>(v[k]);
}
template <int NV>
__device__ inline void wave_sums_i32_to_f64(const int (&v)[NV], double (&out)[NV]) {
    int p[2 * NV];
#pragma unroll
    for (int k = 0; k < NV; k++) { p[2 * k] = v[k] & 0xFFFF; p[2 * k + 1] = v[k] >> 16; }
    dpp_add_all<0xB1>(p); dpp_add_all<0x4E>(p); dpp_add_all<0x141>(p); dpp_add_all<0x140>(p);
#pragma unroll
    for (int k = 0; k < 2 * NV; k++) p[k] += __builtin_amdgcn_update_dpp(0, p[k], 0x142, 0xA, 0xF, false);   // row_bcast:15
#pragma unroll
    for (int k = 0; k < 2 * NV; k++) p[k] += __builtin_amdgcn_update_dpp(0, p[k], 0x143, 0xC, 0xF, false);   // row_bcast:31
#pragma unroll
    for (int k = 0; k < NV; k++)
        out[k] = (double)__builtin_amdgcn_readlane(p[2 * k + 1], 63) * 65536.0 + (double)__builtin_amdgcn_readlane(p[2 * k], 63);
}
// the same over the workgroup (waves combined through LDS); same value in every thread
template <int NV>
__device__ inline void klt_block_sum(const int (&v)[NV], double (&out)[NV], double (*s_part)[8]) {
    const int wave = threadIdx.x >> 6;
    double w[NV];
    wave_sums_i32_to_f64<NV>(v, w);
#pragma unroll
    for (int k = 0; k < NV; k++) {
        if (KLT_WAVES == 1) out[k] = w[k];
        else if ((threadIdx.x & 63) == 0) s_part[wave][k] = w[k];
    }
    if (KLT_WAVES == 1) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double t = 0;
#pragma unroll
        for (int ww = 0; ww < KLT_WAVES; ww++) t += s_part[ww][k];
        out[k] = t;
    }
    __syncthreads();
}

// every product in the window loops has operands below 2^23 in magnitude: full-rate 24-bit multiplies
#define M24(a, b) __mul24((a), (b))

__device__ inline int cv_round(float v) { return (int)rintf(v); }
__device__ inline int cv_floor(float v) { return (int)floorf(v); }

struct LkWeights { v2s top, bot; };

// one keypoint's template of one pyramid level in HBM (KfDev::tmpl): TQ x 64 uint4 (lane-major:
// q * 64 + lane, a coalesced 1 KB per load instruction) = the NPAIR int16 pairs of I, Ix, Iy of every lane,
// then this header
enum { KLT_OUTSIDE = 0, KLT_FLAT = 1, KLT_TRACK = 2 };    // reference window outside the image / minEig below the threshold
struct KltTmplHeader { int state; float A11, A12, A22; double cI1, cI2; };
template <int NPAIR>
struct KltTmpl {
    static constexpr int TQ = (3 * NPAIR + 3) / 4;
    static constexpr int BYTES = TQ * 64 * 16 + 64;
};
// the four 14-bit bilinear weights of calcOpticalFlowPyrLK for the fractions (fa, fb)
__device__ inline LkWeights lk_weights(float fa, float fb) {
    const int W_BITS = 14;
    const int iw00 = cv_round((1.f - fa) * (1.f - fb) * (1 << W_BITS));
    const int iw01 = cv_round(fa * (1.f - fb) * (1 << W_BITS));
    const int iw10 = cv_round((1.f - fa) * fb * (1 << W_BITS));
    const int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    return LkWeights{as_v2s(pack16(iw00, iw01)), as_v2s(pack16(iw10, iw11))};
}

// stage rows [y0, y0+rows) x columns [x0, x0+4*nq) of `im` into an LDS tile with row stride `ls`
// (x0 % 4 == 0). Dword loads when the rectangle lies inside the image and the image is dword
// aligned, else one byte per (row, column) with BORDER_REFLECT_101 addressing.
template <int LS, int MAXROWS>
__device__ inline void stage_tile(uint8_t* tile, const ImgView& im, int x0, int y0, int nq, int rows) {
    const int tid = threadIdx.x;
    const bool fast = x0 >= 0 && y0 >= 0 && x0 + 4 * nq <= im.w && y0 + rows <= im.h &&
                      (((reinterpret_cast<uintptr_t>(im.data) | (uintptr_t)im.stride) & 3) == 0);
    if (fast) {
        constexpr int RPP = KLT_THREADS / 16;                        // rows per pass, 16 dword lanes per row
        constexpr int NP = (MAXROWS + RPP - 1) / RPP;
        const int c4 = tid & 15, r0 = tid >> 4;
        if (c4 < nq) {
            const uint8_t* g = im.g() + M24(y0 + r0, im.stride) + x0 + 4 * c4;
            uint32_t v[NP];
#pragma unroll
            for (int u = 0; u < NP; u++)
                v[u] = (r0 + u * RPP < rows) ? *reinterpret_cast<const uint32_t*>(g + M24(u * RPP, im.stride)) : 0u;
#pragma unroll
            for (int u = 0; u < NP; u++)
                if (r0 + u * RPP < rows) *reinterpret_cast<uint32_t*>(&tile[(r0 + u * RPP) * LS + 4 * c4]) = v[u];
        }
    } else {
        const int tcol = tid & 63, trow0 = tid >> 6;
        if (tcol < 4 * nq) {
            const int gx = reflect101(x0 + tcol, im.w);
            for (int r = trow0 * 8; r < rows; r += 8 * KLT_WAVES) {      // 8 independent loads in flight
                uint8_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int gy = reflect101(y0 + min(r + u, rows - 1), im.h);
                    v[u] = im.g()[M24(gy, im.stride) + gx];            // offsets < 2^31
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (r + u < rows) tile[(r + u) * LS + tcol] = v[u];
            }
        }
    }
}

// Search tile: rows [y0, y0+rows) x columns [x0, x0+4*nq) of `im` as (pixel << 7) 16-bit values in
// two copies, copy 0 from column 0 and copy 1 from column 1 (GEO::J2S dwords per row, GEO::J2COPY
// dwords apart). Inside the image: one dword per lane, 16 lanes per row (a DPP row: the next
// dword's first pixel comes from the neighbouring lane); else byte by byte with BORDER_REFLECT_101.
// The in-image case in two halves, so that a caller can have the loads in flight while it does something else.
__device__ inline bool tile_in_image(const ImgView& im, int x0, int y0, int nq, int rows) {
    return x0 >= 0 && y0 >= 0 && x0 + 4 * nq <= im.w && y0 + rows <= im.h &&
           (((reinterpret_cast<uintptr_t>(im.data) | (uintptr_t)im.stride) & 3) == 0);
}
template <class GEO>
struct KltTileRegs {
    static constexpr int RPP = KLT_THREADS / 16;                     // rows per pass
    static constexpr int NP = (GEO::TJROWS + RPP - 1) / RPP;
    uint32_t v[NP];
};
template <class GEO>
__device__ inline void tile_j2_load(KltTileRegs<GEO>& t, const ImgView& im, int x0, int y0, int nq, int rows) {
    constexpr int RPP = KltTileRegs<GEO>::RPP, NP = KltTileRegs<GEO>::NP;
    const int c4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    const uint8_t* g = im.g() + M24(y0 + r0, im.stride) + x0 + 4 * min(c4, nq - 1);
#pragma unroll
    for (int u = 0; u < NP; u++)
        t.v[u] = (r0 + u * RPP < rows) ? *reinterpret_cast<const uint32_t*>(g + M24(u * RPP, im.stride)) : 0u;
}
template <class GEO>
__device__ inline void tile_j2_store(uint32_t* tile, const KltTileRegs<GEO>& tr, int nq, int rows) {
    constexpr int RPP = KltTileRegs<GEO>::RPP, NP = KltTileRegs<GEO>::NP, KLT_J2S = GEO::J2S, KLT_J2COPY = GEO::J2COPY;
    const int c4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    SVO_LDS(uint32_t)* t0 = (SVO_LDS(uint32_t)*)tile + r0 * KLT_J2S + 2 * c4;
#pragma unroll
    for (int u = 0; u < NP; u++) {
        const uint32_t v = tr.v[u];
        const uint32_t e7 = (v & 0x00ff00ffu) << 7, o7 = ((v >> 8) & 0x00ff00ffu) << 7;   // (p0, p2), (p1, p3), each << 7
        const uint32_t ne7 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)e7, 0x101 /* row_shl:1 */, 0xF, 0xF, false);
        if (c4 < nq && r0 + u * RPP < rows) {
            SVO_LDS(uint32_t)* t = t0 + u * RPP * KLT_J2S;
            t[0] = __builtin_amdgcn_perm(o7, e7, 0x05040100u);                 // (p0, p1)
            t[1] = __builtin_amdgcn_perm(o7, e7, 0x07060302u);                 // (p2, p3)
            t[KLT_J2COPY] = __builtin_amdgcn_perm(e7, o7, 0x07060100u);        // (p1, p2)
            t[KLT_J2COPY + 1] = __builtin_amdgcn_perm(ne7, o7, 0x05040302u);   // (p3, p4): p4 from the next lane (unused in the last column)
        }
    }
}
template <class GEO>
__device__ inline void stage_tile_j2(uint32_t* tile, const ImgView& im, int x0, int y0, int nq, int rows) {
    constexpr int KLT_J2S = GEO::J2S, KLT_J2COPY = GEO::J2COPY;
    const int tid = threadIdx.x;
    if (tile_in_image(im, x0, y0, nq, rows)) {
        KltTileRegs<GEO> tr;
        tile_j2_load<GEO>(tr, im, x0, y0, nq, rows);
        tile_j2_store<GEO>(tile, tr, nq, rows);
    } else {
        const int tcol = tid & 63, trow0 = tid >> 6;
        SVO_LDS(uint16_t)* t16 = (SVO_LDS(uint16_t)*)tile;
        if (tcol < 4 * nq) {
            const int gx = reflect101(x0 + tcol, im.w);
            for (int r = trow0 * 8; r < rows; r += 8 * KLT_WAVES) {      // 8 independent loads in flight
                uint8_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int gy = reflect101(y0 + min(r + u, rows - 1), im.h);
                    v[u] = im.g()[M24(gy, im.stride) + gx];            // offsets < 2^31
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (r + u < rows) {
                        const uint16_t pv = (uint16_t)((int)v[u] << 7);
                        t16[(r + u) * 2 * KLT_J2S + tcol] = pv;
                        if (tcol > 0) t16[2 * KLT_J2COPY + (r + u) * 2 * KLT_J2S + tcol - 1] = pv;
                    }
            }
        }
    }
}

// CW: threads per window row: 32 when w+1 <= 32 (2 row groups of 16 rows), else 36 (one group of 36
// rows, 36 of the 64 threads). A thread owns column lc and the RPT consecutive rows from lr*RPT.
// One wavefront per keypoint (KLT_THREADS = 64): no cross-wave barrier in the iteration, and a
// single-wave workgroup finds a slot while the Gauss-Newton kernels of other sequence groups hold
// most of a CU's registers (+3 % frames/s against 128 threads on the 768-sequence bench).
template <int CW>
__device__ __forceinline__ void klt_track_body(const KltArgs* __restrict__ args) {
    constexpr int NG = KLT_THREADS / CW;                               // row groups
    constexpr int ROWS = CW;                                            // tap rows to cover (w+1 <= CW)
    constexpr int RPT = (((ROWS + NG - 1) / NG) + 1) & ~1;              // even: rows are kept as pairs
    constexpr int NPAIR = RPT / 2;
    const KltArgs& a = args[blockIdx.y];
    const int n = *G(a.n_ptr);
    const int kp = blockIdx.x;
    if (kp >= n) return;
    const int tid = threadIdx.x;
#ifdef SVO_KLT_PHASES
    unsigned long long t_phase_ = __builtin_amdgcn_s_memtime();
    KLT_COUNT(8, 1);
#endif
    const int win = a.win;
    const int RW = win + 3, DW = win + 1, TJ = DW + 2 * KLT_MARGIN;
    const int lr = tid / CW, lc = tid - lr * CW;
    const bool row_on = lr < NG;                 // CW = 36: the last 20 threads only help with the tiles
    const int y0 = row_on ? lr * RPT : 0;

    // one LDS block: while the template is built, s_d (packed (dx, dy) int16 at the tap positions + one
    // slack row) and behind it the reference tile s_I; afterwards the search tile (two 16-bit copies)
    using GEO = KltGeom<CW>;
    constexpr int KLT_DW = GEO::DW, KLT_RS = GEO::RS, KLT_RROWS = GEO::RROWS, KLT_J2S = GEO::J2S, KLT_J2COPY = GEO::J2COPY;
    __shared__ __attribute__((aligned(16))) int s_mem[GEO::LDS_DWORDS];
    int* const s_d = s_mem;
    uint8_t* const s_I = reinterpret_cast<uint8_t*>(s_mem + GEO::SD);
    uint32_t* const s_J2 = reinterpret_cast<uint32_t*>(s_mem);
    __shared__ double s_part[KLT_WAVES][8];

    // the keyframe record is wave-uniform: its fields are read where they are used. (A local copy of
    // the struct lands in scratch memory — 216 B per lane written and read back per keypoint, 0.7 GB
    // of HBM writes per 256-sequence launch — because lk[] is indexed with the run-time level.)
    const int kfid = __builtin_amdgcn_readfirstlane(a.kf_id ? G(a.kf_id)[kp] : 0);
    SVO_GP(const KfDev) kfp = G(a.kfs) + kfid;

    svo_kp2d ref;
    float nx, ny;
    if (a.proj_pose) {
        // frame.kps.kps2d = project_keypoints(estimated_pose, kps3d)   (stereo_slam.cpp:73-80)
        const CamD camd = make_camd(a.cam.fx, a.cam.fy, a.cam.cx, a.cam.cy, a.cam);
        svo_kp2d q;
        if (a.proj_mats) {
            const PoseMats pm = *G(a.proj_mats);
            q = project_point(pm.Rd, pm.t, camd, G(a.kps3d)[kp]);
        } else {
            float pose[6];
            for (int i = 0; i < 6; i++) pose[i] = G(a.proj_pose)[i];
            PoseMats pm;
            pose_mats(pose, pm);
            q = project_point(pm.Rd, pm.t, camd, G(a.kps3d)[kp]);
        }
        nx = q.x; ny = q.y;
        ref = G(kfp->kps2d)[G(a.kp_index)[kp]];
        if (tid == 0) {
            G(a.proj_out)[kp] = q;
            if (a.ref_out) G(a.ref_out)[kp] = ref;
        }
    } else {
        ref = G(a.prev_pts)[kp];
        const svo_kp2d q = G(a.cur_pts)[kp];
        nx = q.x; ny = q.y;
    }

    // template cache of the keyframe (tracker only; null: templates are built every frame)
    const uint8_t* tmpl_base = (a.proj_pose && kfp->tmpl_win == a.win) ? (const uint8_t*)kfp->tmpl : nullptr;
    const int tmpl_cap = kfp->tmpl_cap;
    // (read here, with the record's other fields, not per level: the keyframe table is touched sparsely, and a scalar
    // load from it in every level was a round trip to HBM of ~3 K cycles each, profiles/r03_klt_phases.txt)
    SVO_GP(uint8_t) tmpl_valid = G(kfp->tmpl_valid);
    const int kpi = a.proj_pose ? __builtin_amdgcn_readfirstlane(G(a.kp_index)[kp]) : 0;
    const int maxLevel = min(kfp->n_lk, a.n_cur) - 1;
    const float halfWin = (win - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    double epsilon = 0.01;
    epsilon *= epsilon;
    int status = 1;
    float err = 0;
    const bool col_on = row_on && lc < win;

    KLT_PHASE(0);                                  // prologue: arguments, keyframe record, projection
    for (int level = maxLevel; level >= 0; level--) {
        const ImgView J = a.cur[level];
        const float lscale = (float)(1. / (1 << level));
        float prevx = ref.x * lscale, prevy = ref.y * lscale;
        if (level == maxLevel) { nx = nx * lscale; ny = ny * lscale; }
        else { nx = nx * 2.f; ny = ny * 2.f; }
        float nextx = nx, nexty = ny;

        // ---- the template of the level: I, Ix, Iy of the reference window at its sub-pixel position (the
        // thread's rows, int16 pairs in registers), the covariance matrix of the derivatives and the sums
        // sum I Ix / sum I Iy. They depend on the KEYFRAME only (its pyramid and the keypoint's position
        // there never change), so a tracker keeps them in HBM from the first frame that builds them
        // (KfDev::tmpl, 6 KB per keypoint and level — 288 GB is what makes that a cheap trade) and later
        // frames load them: 6 x 16 B per lane instead of ~1000 VALU + 450 scalar instructions of tile
        // staging, Scharr, interpolation and window sums per level. The same values either way.
        int tIw[NPAIR], tIx[NPAIR], tIy[NPAIR];
        float A11 = 0, A12 = 0, A22 = 0;
        double cI1 = 0, cI2 = 0;
        int lstate = KLT_TRACK;
        LkWeights wt;
        SVO_GP(uint4) rec = nullptr;
        SVO_GP(uint8_t) vflag = nullptr;
        if (tmpl_base && kpi < tmpl_cap) {
            rec = (SVO_GP(uint4))(tmpl_base + ((size_t)kpi * SVO_LK_LEVELS + level) * KltTmpl<NPAIR>::BYTES);
            vflag = tmpl_valid + kpi * SVO_LK_LEVELS + level;
        }
        KLT_PHASE(5);                              // (diagnostic split of phase 1: image views and addresses of the level)
        // A cached level costs memory round trips, not arithmetic (profiles/r03_klt_phases.txt: a third of a
        // wavefront's life): the "stored" flag, the header, the template and the search tile around the predicted
        // position do not depend on each other's values, so all four are requested here, before the first is
        // looked at (until then: flag, then header, then template + tile — three dependent round trips per level).
        // What a miss does not need is dropped. Alone the kernel takes as long as before — it is not waiting for
        // memory at 16 wavefronts per CU — but its wavefronts hold their registers and LDS for less time: +0.8 %
        // frames/s in the mix (eight interleaved pairs, profiles/r03_ab_steps.txt).
        const int TW = (TJ + 3 + 3) & ~3;            // tile columns: covers the margin from any dword phase
        int tx0 = 0, ty0 = 0;
        bool have_tile = false, tile_regs = false;
        KltTileRegs<GEO> pre;
        uint4 hq0 = make_uint4(0, 0, 0, 0), hq1 = hq0;
        uint4 tq[KltTmpl<NPAIR>::TQ];
        int flag = 0;
        if (vflag) {
            flag = *vflag;
            SVO_GP(const uint4) hd = rec + KltTmpl<NPAIR>::TQ * 64;
            hq0 = hd[0]; hq1 = hd[1];
#pragma unroll
            for (int q = 0; q < KltTmpl<NPAIR>::TQ; q++) tq[q] = rec[q * 64 + tid];
            const int px = cv_floor(nextx - halfWin), py = cv_floor(nexty - halfWin);      // where iteration 0 will look
            const int qx0 = (px - KLT_MARGIN) & ~3, qy0 = py - KLT_MARGIN;
            if (tile_in_image(J, qx0, qy0, TW >> 2, TJ)) {
                tile_j2_load<GEO>(pre, J, qx0, qy0, TW >> 2, TJ);
                tx0 = qx0; ty0 = qy0; tile_regs = true;
            }
        }
        KLT_PHASE_WAIT(6);                         // (diagnostic builds: everything requested above has arrived)
        const bool hit = flag != 0;
        if (hit) {
            static_assert(sizeof(KltTmplHeader) == 32, "two 16-byte loads");
            lstate = (int)hq0.x; A11 = __uint_as_float(hq0.y); A12 = __uint_as_float(hq0.z); A22 = __uint_as_float(hq0.w);
            cI1 = __builtin_bit_cast(double, ((unsigned long long)hq1.y << 32) | hq1.x);
            cI2 = __builtin_bit_cast(double, ((unsigned long long)hq1.w << 32) | hq1.z);
            if (lstate == KLT_TRACK) {
                const uint32_t* f = reinterpret_cast<const uint32_t*>(tq);
#pragma unroll
                for (int k = 0; k < NPAIR; k++) { tIw[k] = (int)f[k]; tIx[k] = (int)f[NPAIR + k]; tIy[k] = (int)f[2 * NPAIR + k]; }
            }
        } else {
        tile_regs = false;                           // (the template is built in the tile's LDS: stage it afterwards)
        // (this path runs once per keypoint and keyframe; opaque copies of the lane's column and first row keep the
        // compiler from computing its sixteen row addresses before the level loop and holding them through the hot path)
        int lc_b = lc, y0_b = y0;
        asm volatile("" : "+v"(lc_b), "+v"(y0_b));
        const ImgView I = kfp->lk[level];            // (the keyframe's image: only a template that is built needs it)
        prevx -= halfWin; prevy -= halfWin;
        const int iprevx = cv_floor(prevx), iprevy = cv_floor(prevy);
        if (iprevx < -win || iprevx >= I.w || iprevy < -win || iprevy >= I.h) {
            lstate = KLT_OUTSIDE;
        } else {
        wt = lk_weights(prevx - iprevx, prevy - iprevy);

        __syncthreads();
        // (w+3)^2 tile of I around the window: rows iprevy-1 .., columns from the dword boundary left of iprevx-1
        const int ix0 = (iprevx - 1) & ~3;
        const int ox = iprevx - 1 - ix0;                 // 0..3
        stage_tile<KLT_RS, KLT_RROWS>(s_I, I, ix0, iprevy - 1, (ox + RW + 3) >> 2, RW);
        __syncthreads();

        // Scharr (calcSharrDeriv) at the (w+1)^2 tap positions, sliding down the thread's rows:
        // hd = p[+1]-p[-1], hs = 3(p[-1]+p[+1]) + 10 p[0] per tile row; dx = 3(hd_0+hd_2) + 10 hd_1,
        // dy = hs_2 - hs_0. Zero outside the image.
        if (lc_b < DW && row_on) {
            const int gx = iprevx + lc_b;
            const bool xin = (unsigned)gx < (unsigned)I.w;
            const uint32_t pcol = lds_addr(&s_I[y0_b * KLT_RS + ox + lc_b]);
            int p02[RPT + 2], p1[RPT + 2];
            lds_col_nowait<KLT_RS, 1>(pcol, p1);
            lds_pairs<KLT_RS, 0, 2>(pcol, p02);
            lds_fence(p1);
            const v2s kd = as_v2s(pack16(-1, 1)), ks = as_v2s(pack16(3, 3));
            int hd0 = dot2(as_v2s(p02[0]), kd, 0), hs0 = dot2(as_v2s(p02[0]), ks, M24(p1[0], 10));
            int hd1 = dot2(as_v2s(p02[1]), kd, 0), hs1 = dot2(as_v2s(p02[1]), ks, M24(p1[1], 10));
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                const int r = y0_b + u;
                const int hd2 = dot2(as_v2s(p02[u + 2]), kd, 0), hs2 = dot2(as_v2s(p02[u + 2]), ks, M24(p1[u + 2], 10));
                const int dx = M24(hd0 + hd2, 3) + M24(hd1, 10);
                const int dy = hs2 - hs0;
                if (r < DW) s_d[r * KLT_DW + lc_b] = (xin && (unsigned)(iprevy + r) < (unsigned)I.h) ? pack16(dx, dy) : 0;
                hd0 = hd1; hd1 = hd2; hs0 = hs1; hs1 = hs2;
            }
        }
        __syncthreads();

        // template of the thread's rows (registers) + covariance of the derivatives
        // (per-thread int32 partials: <= 36 pixels, each product < 2^24)
        int a11 = 0, a12 = 0, a22 = 0, c1 = 0, c2 = 0;
        {
            const int lcs = col_on ? lc_b : 0;                             // keep idle columns in bounds
            // tile row y+1 holds image row iprevy+y
            int ipr[RPT + 1];
            lds_pairs<KLT_RS, 0, 1>(lds_addr(&s_I[(y0_b + 1) * KLT_RS + ox + lcs + 1]), ipr);
            const int* dcol = &s_d[y0_b * KLT_DW + lcs];
            int d0 = dcol[0], d1 = dcol[1];
            v2s dxp = as_v2s((int)__builtin_amdgcn_perm((uint32_t)d1, (uint32_t)d0, 0x05040100u));
            v2s dyp = as_v2s((int)__builtin_amdgcn_perm((uint32_t)d1, (uint32_t)d0, 0x07060302u));
            int iv[2], ixv[2], iyv[2];
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                const int y = y0_b + u;
                d0 = dcol[(u + 1) * KLT_DW]; d1 = dcol[(u + 1) * KLT_DW + 1];
                const v2s dxp1 = as_v2s((int)__builtin_amdgcn_perm((uint32_t)d1, (uint32_t)d0, 0x05040100u));
                const v2s dyp1 = as_v2s((int)__builtin_amdgcn_perm((uint32_t)d1, (uint32_t)d0, 0x07060302u));
                int ival = dot2(as_v2s(ipr[u]), wt.top, dot2(as_v2s(ipr[u + 1]), wt.bot, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                int ixval = dot2(dxp, wt.top, dot2(dxp1, wt.bot, 1 << (W_BITS - 1))) >> W_BITS;
                int iyval = dot2(dyp, wt.top, dot2(dyp1, wt.bot, 1 << (W_BITS - 1))) >> W_BITS;
                if (!(col_on && y < win)) { ival = 0; ixval = 0; iyval = 0; }
                a11 += M24(ixval, ixval); a12 += M24(ixval, iyval); a22 += M24(iyval, iyval);
                c1 += M24(ival, ixval); c2 += M24(ival, iyval);      // (|I| < 2^13, |dI| < 2^12, 18 rows: below 2^30)
                iv[u & 1] = ival; ixv[u & 1] = ixval; iyv[u & 1] = iyval;
                if (u & 1) {
                    tIw[u >> 1] = pack16(iv[0], iv[1]);
                    tIx[u >> 1] = pack16(ixv[0], ixv[1]);
                    tIy[u >> 1] = pack16(iyv[0], iyv[1]);
                }
                dxp = dxp1; dyp = dyp1;
            }
        }
        double sA[5];
        {
            const int pa[5] = {a11, a12, a22, c1, c2};
            klt_block_sum<5>(pa, sA, s_part);
        }
        cI1 = sA[3]; cI2 = sA[4];                                 // sum I Ix, sum I Iy over the window
        A11 = (float)sA[0] * FLT_SCALE; A12 = (float)sA[1] * FLT_SCALE; A22 = (float)sA[2] * FLT_SCALE;
        const float D0 = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (2 * win * win);
        if ((double)minEig < 1e-4 || D0 < FLT_EPSILON) lstate = KLT_FLAT;
        }   // reference window inside the image
        if (vflag) {                       // first frame that sees this keypoint at this level: keep the template
            if (lstate == KLT_TRACK) {
                uint4 v[KltTmpl<NPAIR>::TQ];
                uint32_t* f = reinterpret_cast<uint32_t*>(v);
#pragma unroll
                for (int k = 0; k < 4 * KltTmpl<NPAIR>::TQ; k++) f[k] = 0;
#pragma unroll
                for (int k = 0; k < NPAIR; k++) { f[k] = (uint32_t)tIw[k]; f[NPAIR + k] = (uint32_t)tIx[k]; f[2 * NPAIR + k] = (uint32_t)tIy[k]; }
#pragma unroll
                for (int q = 0; q < KltTmpl<NPAIR>::TQ; q++) rec[q * 64 + tid] = v[q];
            }
            if (tid == 0) {
                SVO_GP(KltTmplHeader) hd = (SVO_GP(KltTmplHeader))(rec + KltTmpl<NPAIR>::TQ * 64);
                hd->state = lstate; hd->A11 = A11; hd->A12 = A12; hd->A22 = A22; hd->cI1 = cI1; hd->cI2 = cI2;
                *vflag = 1;                // (read by later launches only: ordered by the kernel boundary)
            }
        }
        }   // template built here
        KLT_PHASE(1);                              // template of the level requested (cache) or built
        if (lstate == KLT_OUTSIDE) {
            if (level == 0) { status = 0; err = 0; }
            continue;
        }
        if (lstate == KLT_FLAT) {
            if (level == 0) status = 0;
            continue;
        }
        float D = A11 * A22 - A12 * A12;
        D = 1.f / D;
        nextx -= halfWin; nexty -= halfWin;
        float prevDx = 0, prevDy = 0;

        // stage the search tile [tx0, tx0+TW) x [ty0, ty0+TJ) around the window at (cx, cy)
        auto load_tile = [&](int cx, int cy) {
            tx0 = (cx - KLT_MARGIN) & ~3; ty0 = cy - KLT_MARGIN;
            __syncthreads();
            KLT_PHASE_WAIT(12);                    // (diagnostic: everything requested before the tile has arrived)
            stage_tile_j2<GEO>(s_J2, J, tx0, ty0, TW >> 2, TJ);
            KLT_PHASE_WAIT(13);                    // (... the tile's loads and its LDS stores)
            __syncthreads();
            have_tile = true;
            KLT_COUNT(9, 1);
            KLT_PHASE(2);                          // search tile staged (+ whatever loads were still in flight)
        };
        // J(x+d) of the thread's row pair k (5 fractional bits, like the template) for the window whose
        // rows were read into jr: the tile holds pixel << 7, so (sum + 2^8) >> 9 is the high half of
        // sum * 2^7 + 2^15, and the pair is the two high halves
        const int round_half = __builtin_amdgcn_readfirstlane(1 << (W_BITS - 5 - 1 + 7));
        auto row_pair = [&](const int (&jr)[RPT + 1], const LkWeights& w, int k) -> v2s {
            const int d0 = dot2(as_v2s(jr[2 * k]), w.top, dot2_uc(as_v2s(jr[2 * k + 1]), w.bot, round_half));
            const int d1 = dot2(as_v2s(jr[2 * k + 1]), w.top, dot2_uc(as_v2s(jr[2 * k + 2]), w.bot, round_half));
            return as_v2s((int)__builtin_amdgcn_perm((uint32_t)d1, (uint32_t)d0, 0x07060302u));
        };
        // the bilinear pairs (p[x], p[x+1]) of the thread's column and rows for the window at tile offset (wx, wy)
        auto load_pairs = [&](int wx, int wy, int (&jr)[RPT + 1]) {
            const int x = wx + lc;
            const SVO_LDS(uint32_t)* q = (const SVO_LDS(uint32_t)*)s_J2 + (x & 1) * KLT_J2COPY + (wy + y0) * KLT_J2S + (x >> 1);
#pragma unroll
            for (int u = 0; u <= RPT; u++) jr[u] = (int)q[u * KLT_J2S];
        };

        if (tile_regs) {                             // the tile requested at the level's start
            __syncthreads();
            tile_j2_store<GEO>(s_J2, pre, TW >> 2, TJ);
            __syncthreads();
            have_tile = true;
            KLT_COUNT(9, 1);
            KLT_PHASE(2);
        }
#ifndef SVO_KLT_MAXIT
#define SVO_KLT_MAXIT 30     /* (diagnostic builds count the instructions outside the iteration with 0) */
#endif
        for (int j = 0; j < SVO_KLT_MAXIT; j++) {
            KLT_COUNT(10, 1);
            const int inextx = cv_floor(nextx), inexty = cv_floor(nexty);
            if (inextx < -win || inextx >= J.w || inexty < -win || inexty >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            if (!have_tile || inextx < tx0 || inexty < ty0 || inextx + DW > tx0 + TW || inexty + DW > ty0 + TJ)
                load_tile(inextx, inexty);
            wt = lk_weights(nextx - inextx, nexty - inexty);
            // b = sum (J - I) dI = sum J dI - sum I dI: the second sum is a constant of the level (cI1, cI2)
            int b1 = 0, b2 = 0;
            if (col_on) {
                int jr[RPT + 1];
                load_pairs(inextx - tx0, inexty - ty0, jr);
#pragma unroll
                for (int k = 0; k < NPAIR; k++) {
#ifdef SVO_KLT_NOFOLD
                    const v2s jv = row_pair(jr, wt, k) - as_v2s(tIw[k]);
#else
                    const v2s jv = row_pair(jr, wt, k);
#endif
                    b1 = dot2(jv, as_v2s(tIx[k]), b1);        // rows outside the window have Ix = Iy = 0
                    b2 = dot2(jv, as_v2s(tIy[k]), b2);
                }
            }
            double sB[2];
            {
                const int pb[2] = {b1, b2};
                klt_block_sum<2>(pb, sB, s_part);
            }
#ifdef SVO_KLT_NOFOLD
            const float fb1 = (float)sB[0] * FLT_SCALE, fb2 = (float)sB[1] * FLT_SCALE;
            (void)cI1; (void)cI2;
#else
            const float fb1 = (float)(sB[0] - cI1) * FLT_SCALE, fb2 = (float)(sB[1] - cI2) * FLT_SCALE;
#endif
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
        KLT_PHASE(3);                              // iterations

        if (status && level == 0) {
            const float npx = nx - halfWin, npy = ny - halfWin;
            const int inx = cv_floor(npx), iny = cv_floor(npy);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                status = 0;
                continue;
            }
            if (!have_tile || inx < tx0 || iny < ty0 || inx + DW > tx0 + TW || iny + DW > ty0 + TJ)
                load_tile(inx, iny);
            wt = lk_weights(npx - inx, npy - iny);
            int e = 0;
            if (col_on) {
                int jr[RPT + 1];
                load_pairs(inx - tx0, iny - ty0, jr);
#pragma unroll
                for (int k = 0; k < NPAIR; k++) {
                    const v2s diff = row_pair(jr, wt, k) - as_v2s(tIw[k]);
                    if (y0 + 2 * k < win) e += abs((int)diff.x);
                    if (y0 + 2 * k + 1 < win) e += abs((int)diff.y);
                }
            }
            // < 2^24: the float sum of |diff| is exact in any order
            double sE[1];
            {
                const int pe[1] = {e};
                klt_block_sum<1>(pe, sE, s_part);
            }
            const float errval = (float)sE[0];
            err = errval * 1.f / (32 * win * win);
        }
    }

    KLT_PHASE(4);                                  // error of the final position
    if (tid == 0) {
        G(a.cur_pts)[kp] = svo_kp2d{nx, ny};
        G(a.status)[kp] = (uint8_t)status;
        G(a.err)[kp] = status ? err : INFINITY;   // optical_flow.cpp:46-50
    }
}

// The two kernel shapes. Registers per wavefront decide how many share a SIMD (512 / n): the 32-column shape
// is held to 128 (four; a few registers of the template-building path — run once per keypoint and keyframe — go
// to scratch), the 36-column shape to 256.
template <int CW>
__global__ void klt_track_kernel(const KltArgs* __restrict__ args);
template <>
__global__ __launch_bounds__(KLT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void klt_track_kernel<32>(const KltArgs* __restrict__ args) {
    klt_track_body<32>(args);
}
template <>
__global__ __launch_bounds__(KLT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void klt_track_kernel<36>(const KltArgs* __restrict__ args) {
    klt_track_body<36>(args);
}

#ifdef SVO_KLT_PHASES
}  // namespace svo
// out[0..4]: cycles in prologue / template / tile staging / iterations / error pass, [8] wavefronts,
// [9] tile stagings, [10] iterations; clears the counters
extern "C" int svo_debug_klt_phases(unsigned long long* out16) {
    static unsigned long long host[1024][16];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(svo::g_klt_phases), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 16; i++) {
        out16[i] = 0;
        for (int s = 0; s < 1024; s++) out16[i] += host[s][i];
    }
    memset(host, 0, sizeof(host));
    return hipMemcpyToSymbol(HIP_SYMBOL(svo::g_klt_phases), host, sizeof(host)) == hipSuccess ? 0 : -1;
}
namespace svo {
#endif

// bytes of one keypoint's template of one level for window `win` (KfDev::tmpl)
size_t klt_template_bytes(int win) {
    return win + 1 <= 32 ? KltTmpl<8>::BYTES : KltTmpl<18>::BYTES;
}

void launch_klt(const KltArgs* d_args, int batch, int max_n, int win, hipStream_t stream) {
    if (max_n <= 0) return;
    if (win + 1 <= 32)
        hipLaunchKernelGGL(klt_track_kernel<32>, dim3(max_n, batch), dim3(KLT_THREADS), 0, stream, d_args);
    else
        hipLaunchKernelGGL(klt_track_kernel<36>, dim3(max_n, batch), dim3(KLT_THREADS), 0, stream, d_args);
}

}  // namespace svo
