// Development microbenchmark (not part of the product): what the memory system gives a kernel that moves SHADE's bytes in SHADE's pattern and
// computes nothing.  One launch over a 57 M-slot pool laid out like the pipeline's ([64-slot block][24 rows][64 lanes] of 8-byte cells):
//   prologue   meta word + hit word of the slot (2 rows)
//   state      ray (6 rows), key, unit, hit distance; beta for half the slots; att0 + L for a quarter
//   records    a 160-byte shading record at a random place of a 160 MB array for 45 % of the slots (a triangle hit), one 16-byte texel at a random
//              place of a 100 MB array for half of them (a miss)
//   writes     continuing paths (half): ray, meta, beta or att0; ended paths (half): 24 bytes at samples[unit], then a new sample's ray, key, meta, unit
// The fractions are cfg3's (segments per primary 2.0, 45 % triangle hits); per slot ~190 bytes read and ~110-130 written, as the counters report for
// stream_shade (profiles/r3_cfg3_traffic.json: 186 / 128).  Variants: slots in order or permuted inside their 256-slot block (the material sort), 4 or 8
// waves per SIMD.  hipcc --offload-arch=gfx950 -O3 shade_bound.hip -o shade_bound && ./shade_bound
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define ROWS 24
__device__ __forceinline__ double* cell(double* pool, int f, uint32_t slot) { return pool + ((size_t)(slot >> 6) * ROWS + f) * 64 + (slot & 63u); }
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <bool PERMUTE, int WAVES>
__global__ __launch_bounds__(256, WAVES) void shade_like(double* __restrict__ pool, uint32_t P, const double* __restrict__ recs, uint32_t n_rec,
                                                         const float4* __restrict__ texels, uint32_t n_tex, double* __restrict__ samples, uint32_t n_samples) {
    uint32_t tid = threadIdx.x;
    if (PERMUTE) tid = (tid * 77u + 13u) & 255u;   // a fixed permutation of the block's slots (77 is odd)
    const uint32_t slot = blockIdx.x * 256 + tid;
    if (slot >= P) return;
    // prologue
    const double ma = *cell(pool, 21, slot), ki = *cell(pool, 7, slot);
    const uint32_t h = hash32(slot * 2654435761u + (uint32_t)__double_as_longlong(ma));
    const bool hit_tri = (h % 100u) < 45u, miss = !hit_tri && (h % 100u) < 95u, first = (h >> 8) & 1u, second = (h >> 9) & 1u;
    // state rows, requested together
    double r[6];
#pragma unroll
    for (int k = 0; k < 6; k++) r[k] = *cell(pool, k, slot);
    const double key = *cell(pool, 20, slot), mb = *cell(pool, 22, slot), t = *cell(pool, 6, slot);
    double b[3] = {1, 1, 1}, a0[3] = {0, 0, 0}, L[3] = {0, 0, 0};
    if (!first) {
#pragma unroll
        for (int k = 0; k < 3; k++) b[k] = *cell(pool, 8 + k, slot);
    }
    if (!first && miss && second) {
#pragma unroll
        for (int k = 0; k < 3; k++) { a0[k] = *cell(pool, 14 + k, slot); L[k] = *cell(pool, 11 + k, slot); }
    }
    double acc = ma + ki + key + mb + t + r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + b[0] + b[1] + b[2] + a0[0] + a0[1] + a0[2] + L[0] + L[1] + L[2];
    // the random records
    if (hit_tri) {
        const double2* q = reinterpret_cast<const double2*>(recs + (size_t)(hash32(h) % n_rec) * 20);
#pragma unroll
        for (int k = 0; k < 10; k++) { const double2 v = q[k]; acc += v.x + v.y; }
    } else if (miss) {
        const float4 v = texels[hash32(h ^ 0x9E3779B9u) % n_tex];
        acc += v.x + v.y + v.z;
    }
    // writes
    const bool ended = miss || ((h >> 12) % 10u) == 0u;
    if (!ended) {
#pragma unroll
        for (int k = 0; k < 6; k++) *cell(pool, k, slot) = r[k] + acc;
        *cell(pool, 21, slot) = ma + 1.0;
        if (first) { for (int k = 0; k < 3; k++) *cell(pool, 14 + k, slot) = acc; }
        else { for (int k = 0; k < 3; k++) *cell(pool, 8 + k, slot) = b[k] * acc; }
    } else {
        double* s = samples + (size_t)((slot + (uint32_t)__double_as_longlong(mb)) % n_samples) * 3;   // units are dealt in order: a block's samples lie together
        s[0] = acc; s[1] = acc; s[2] = acc;
#pragma unroll
        for (int k = 0; k < 6; k++) *cell(pool, k, slot) = acc + k;
        *cell(pool, 20, slot) = key + 1.0; *cell(pool, 21, slot) = 0.0; *cell(pool, 22, slot) = mb + 1.0;
    }
}

// the streaming bound for the same byte mix: 10.6 GB read and 7.3 GB written per launch (the counters' figures for stream_shade), perfectly coalesced 16-byte accesses
__global__ __launch_bounds__(256) void stream_mix(const float4* __restrict__ a, size_t n_read, float4* __restrict__ b, size_t n_write) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t k = i; k < n_read; k += stride) { const float4 v = a[k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    for (size_t k = i; k < n_write; k += stride) b[k] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const uint32_t P = 57000000u / 256u * 256u;   // the slots a cfg3 launch carries
    const size_t pool_bytes = (size_t)(P / 64) * ROWS * 64 * 8;
    const uint32_t n_rec = 1000000, n_tex = 4096 * 2048 * 12 / 16, n_samples = 1u << 30;
    double *pool, *recs, *samples; float4* tex;
    CK(hipMalloc(&pool, pool_bytes)); CK(hipMalloc(&recs, (size_t)n_rec * 160)); CK(hipMalloc(&tex, (size_t)n_tex * 16)); CK(hipMalloc(&samples, (size_t)n_samples * 24));
    CK(hipMemset(pool, 0, pool_bytes)); CK(hipMemset(recs, 0, (size_t)n_rec * 160)); CK(hipMemset(tex, 0, (size_t)n_tex * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(P / 256), block(256);
    for (int variant = 0; variant < 4; variant++) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL((shade_like<false, 4>), grid, block, 0, 0, pool, P, recs, n_rec, tex, n_tex, samples, n_samples);
            if (variant == 1) hipLaunchKernelGGL((shade_like<true, 4>), grid, block, 0, 0, pool, P, recs, n_rec, tex, n_tex, samples, n_samples);
            if (variant == 2) hipLaunchKernelGGL((shade_like<false, 8>), grid, block, 0, 0, pool, P, recs, n_rec, tex, n_tex, samples, n_samples);
            if (variant == 3) hipLaunchKernelGGL((shade_like<true, 8>), grid, block, 0, 0, pool, P, recs, n_rec, tex, n_tex, samples, n_samples);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        // bytes by the model above: reads 16 + 72 + 0.5 * 24 + 0.25 * 0.5 * 48 (att0 + L: non-first misses with `second`) + 0.45 * 160 + 0.5 * 16; writes: see kernel
        const double rd = 16 + 72 + 12 + 0.5 * 0.5 * 0.5 * 48 + 0.45 * 160 + 0.5 * 16;
        const double p_end = 0.5 + 0.45 * 0.1 + 0.05 * 0.1, wr = (1 - p_end) * (48 + 8 + 24) + p_end * (24 + 48 + 24);
        printf("%s, %d waves/SIMD: %.3f ms per launch of %u slots  = %.2f TB/s of the modelled %.0f + %.0f bytes per slot\n", (variant & 1) ? "slots permuted in their block" : "slots in order       ", variant < 2 ? 4 : 8, best, P,
               (rd + wr) * P / best / 1e9, rd, wr);
    }
    {
        const size_t n_read = (size_t)10600000000ull / 16, n_write = (size_t)7300000000ull / 16;   // samples[] is 25.8 GB: room for both
        const float4* a = reinterpret_cast<const float4*>(samples); float4* b = reinterpret_cast<float4*>(samples) + n_read;
        for (int blocks : {256 * 8, 256 * 32, 256 * 128}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(stream_mix, dim3(blocks), dim3(256), 0, 0, a, n_read, b, n_write);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("streaming 10.6 GB in + 7.3 GB out, %d blocks: %.3f ms = %.2f TB/s\n", blocks, best, 17.9e9 / best / 1e9);
        }
    }
    return 0;
}
