// svo_reduce.hpp — deterministic workgroup reduction of a small vector of
// float accumulators: wavefront butterfly (64 lanes), one LDS row per wave,
// then one lane per accumulator adds the rows in wave order. The order is
// fixed, so the same inputs always give the same bits.
#pragma once

#include "svo_device.hpp"

namespace svo {

// v[NV] per thread -> sums[0..NV) in LDS, valid for every thread on return.
// red must hold [THREADS/64][32] floats (NV <= 32).
template <int NV, int THREADS>
__device__ inline void block_reduce(float (&v)[NV], float (*red)[32], float* sums) {
    static_assert(NV <= 32, "at most 32 accumulators");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        float x = v[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
        if (lane == 0) red[wave][k] = x;
    }
    __syncthreads();
    if (tid < NV) {
        float s = red[0][tid];
#pragma unroll
        for (int w = 1; w < THREADS / 64; w++) s += red[w][tid];
        sums[tid] = s;
    }
    __syncthreads();
}

}  // namespace svo
