/* zr_rng.h — the RNG *contract* shared by the oracle harness, the CPU restatement and the HIP kernels.
 *
 * The reference draws every random number from one process-global std::mt19937 that is seeded from
 * std::random_device and shared by all render threads (/root/reference/common.hpp:29-34): it has no
 * seed API, so "identical RNG seeds" can only be met by replacing the engine.  This header defines
 * the replacement: a counter-based 64-bit generator (SplitMix64 finaliser over a keyed counter).
 *
 *   main stream   one per (seed, pixel = j*W + i, sample s); draw k = 0,1,2,... in the reference's
 *                 program order: jitter x, jitter y, lens pairs, then per bounce
 *                 the material's scatter draws and the Russian-roulette draw.
 *   medium draw   constant_medium::hit draws *inside* BVH traversal (constant_medium.hpp:64), so the
 *                 number of such draws depends on tree topology.  It is therefore keyed off-stream by
 *                 (main-stream key, bounce index, medium id): every test of one medium within one
 *                 closest-hit query sees the same xi, which makes the result independent of traversal
 *                 order, culling and the reference's duplicated-leaf double test.
 *   scene stream  pixel = 0xFFFFFFFF, sample = stream id: used by scene generators.
 *
 * Mapping to a double follows libstdc++'s std::uniform_real_distribution<double> over a 64-bit URBG
 * (generate_canonical with one engine call): u = double(x) * 2^-64, and a result that rounds up to 1.0
 * is replaced by nextafter(1,0).
 *
 * Plain C99 / C++ / HIP device code.
 */
#ifndef ZR_RNG_H
#define ZR_RNG_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define ZR_HD __host__ __device__ inline
#else
#define ZR_HD static inline
#endif

#define ZR_GOLDEN 0x9E3779B97F4A7C15ull
#define ZR_MEDIUM_TAG 0x4D454449554D5F5Full /* "MEDIUM__" */
#define ZR_SCENE_PIXEL 0xFFFFFFFFull

ZR_HD uint64_t zr_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* key of the main stream of one primary sample */
ZR_HD uint64_t zr_stream_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    return zr_mix64(zr_mix64(seed + ZR_GOLDEN * (pixel + 1ull)) + ZR_GOLDEN * (sample + 1ull));
}

/* k-th 64-bit output (k = 0,1,...) of a main stream */
ZR_HD uint64_t zr_stream_bits(uint64_t key, uint64_t k) {
    return zr_mix64(key + ZR_GOLDEN * (k + 1ull));
}

/* the off-stream draw of medium `medium_id` during closest-hit query number `bounce` (0 = primary) */
ZR_HD uint64_t zr_medium_bits(uint64_t key, uint32_t bounce, uint32_t medium_id) {
    return zr_mix64(zr_mix64(key ^ ZR_MEDIUM_TAG) + ZR_GOLDEN * ((uint64_t)bounce * 65536ull + (uint64_t)medium_id + 1ull));
}

/* libstdc++ generate_canonical<double,53> over one 64-bit word */
ZR_HD double zr_bits_to_unit(uint64_t x) {
    double u = (double)x * 5.42101086242752217003726400434970855712890625e-20; /* 2^-64 */
    return u >= 1.0 ? 0.99999999999999988897769753748434595763683319091796875 : u;
}

#endif /* ZR_RNG_H */
