// synth_render.hip — libsvo_synth.so: ray-casts the seeded synthetic stereo scenes of
// stereo_svo_slam_amd/synth.py (class Scene) on the GPU, one thread per pixel, so that bench.py
// and the long GPU tests get their INPUT frames in seconds instead of minutes of elementwise
// torch launches. Workload generation only: nothing of the hot path lives here, and the product
// library (libsvo_hip.so) does not link it. Same scene model as Scene.render_batch (planes with
// bilinear value-noise textures, nearest hit wins) plus Gaussian sensor noise from a
// counter-based hash; float32 per-pixel math (not bit-identical to the torch path).
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" {

struct svo_synth_plane {
    float p0[3], u[3], v[3];
    float hu, hv;          // half extents; <= 0: unbounded
    float off_u, off_v;    // texture offset in texels
};

struct svo_synth_params {
    svo_synth_plane planes[16];
    int n_planes;
    int tex_size;
    float tpm;             // texels per metre
    float fx, fy, cx, cy;
    int w, h;
    float noise_sigma;
    uint32_t seed;
};

}  // extern "C"

__device__ inline uint32_t pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

// poses: [K][12] = R (row major, 9) + origin (3); out: [K][h][w]
__global__ __launch_bounds__(256) void synth_render_kernel(svo_synth_params p, const float* __restrict__ tex,
                                                           const float* __restrict__ poses,
                                                           uint8_t* __restrict__ out, const uint32_t* __restrict__ img_seed) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int k = blockIdx.z;
    if (x >= p.w || y >= p.h) return;
    const float* R = poses + (size_t)k * 12;
    const float ox = R[9], oy = R[10], oz = R[11];
    const float X = (float)(((double)x - p.cx) / p.fx), Y = (float)(((double)y - p.cy) / p.fy);
    const float dx = R[0] * X + R[1] * Y + R[2], dy = R[3] * X + R[4] * Y + R[5], dz = R[6] * X + R[7] * Y + R[8];
    float best = INFINITY, val = 0.f;
    const int S = p.tex_size;
    for (int i = 0; i < p.n_planes; i++) {
        const svo_synth_plane& pl = p.planes[i];
        const float nx = pl.u[1] * pl.v[2] - pl.u[2] * pl.v[1];
        const float ny = pl.u[2] * pl.v[0] - pl.u[0] * pl.v[2];
        const float nz = pl.u[0] * pl.v[1] - pl.u[1] * pl.v[0];
        const float dn = dx * nx + dy * ny + dz * nz;
        const float num = (pl.p0[0] - ox) * nx + (pl.p0[1] - oy) * ny + (pl.p0[2] - oz) * nz;
        const float s = num / dn;
        if (!(s > 1e-3f) || !(s < best) || !isfinite(s)) continue;
        const float qx = ox + s * dx - pl.p0[0], qy = oy + s * dy - pl.p0[1], qz = oz + s * dz - pl.p0[2];
        const float tu = qx * pl.u[0] + qy * pl.u[1] + qz * pl.u[2];
        const float tv = qx * pl.v[0] + qy * pl.v[1] + qz * pl.v[2];
        if (pl.hu > 0 && !(fabsf(tu) < pl.hu && fabsf(tv) < pl.hv)) continue;
        const float fu = tu * p.tpm + pl.off_u, fv = tv * p.tpm + pl.off_v;
        const float iu = floorf(fu), iv = floorf(fv);
        const float au = fu - iu, av = fv - iv;
        int iu0 = (int)fmodf(iu, (float)S), iv0 = (int)fmodf(iv, (float)S);
        if (iu0 < 0) iu0 += S;
        if (iv0 < 0) iv0 += S;
        const int iu1 = (iu0 + 1) % S, iv1 = (iv0 + 1) % S;
        const float* t = tex + (size_t)i * S * S;
        val = t[iv0 * S + iu0] * (1 - au) * (1 - av) + t[iv0 * S + iu1] * au * (1 - av) +
              t[iv1 * S + iu0] * (1 - au) * av + t[iv1 * S + iu1] * au * av;
        best = s;
    }
    if (p.noise_sigma > 0) {
        const uint32_t h0 = pcg_hash(img_seed[k] ^ pcg_hash((uint32_t)(y * p.w + x) * 2u + p.seed));
        const uint32_t h1 = pcg_hash(h0 + 0x9E3779B9u);
        const float u1 = ((float)(h0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(h1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
        val += p.noise_sigma * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    }
    val = fminf(fmaxf(rintf(val), 0.f), 255.f);
    out[((size_t)k * p.h + y) * p.w + x] = (uint8_t)val;
}

extern "C" int svo_synth_render(const svo_synth_params* p, const float* tex, const float* poses_dev,
                                const uint32_t* img_seed_dev, int K, uint8_t* out, void* stream) {
    if (!p || !tex || !poses_dev || !out || K <= 0 || p->n_planes > 16) return -1;
    dim3 grid((p->w + 63) / 64, (p->h + 3) / 4, K);
    hipLaunchKernelGGL(synth_render_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *p, tex,
                       poses_dev, out, img_seed_dev);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
