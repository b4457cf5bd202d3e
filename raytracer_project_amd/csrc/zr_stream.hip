// zr_stream.hip — render variant 2: streaming wavefront pipeline for gfx950.
//
// The megakernels (variants 0/1) keep a whole path — ray, throughput, radiance, RNG, hit record, traversal — in
// registers: ~256 VGPRs, 2 waves per SIMD, and measured throughput is proportional to the number of resident
// waves (dependent BVH fetches: latency bound).  This variant splits the sample loop into stages so that the
// stage that chases pointers is register-lean and runs at several times the occupancy:
//
//   pool of P path SLOTS resident in HBM (SoA, FP64): ray, throughput, radiance, RNG position, pixel/sample ids
//   round:  EXTEND  persistent waves pull ray indices from a global head (ballot + one atomic per refill), walk the
//                   BVH with the NODE/LEAF wave scheduler (per-lane LDS stack, cull-on-pop) and write (t, object)
//           SHADE   one thread per slot: hit record, emission, scatter, Russian roulette, background; a finished
//                   path adds its radiance to the SLOT's running sum and the slot starts its next sample in place
//                   (regeneration), so every slot carries exactly one segment per round until the frame drains
//   end:    REDUCE  slot sums were written to partial[pixel][lane] when a slot left a pixel; one fixed-order sum
//                   per pixel gives the mean — no atomics on radiance anywhere, the image is bit-reproducible
//
// Work units: unit u = (pixel u / LANES of the frame's pixel list, lane j = u % LANES) = samples j, j + LANES, ... of
// that pixel.  Slot k starts on unit k; a slot that finishes its unit writes its sum to partial[pixel][j] and takes
// the next unit from one of ST_SHARDS interleaved counters (one atomic per wave per round, spread over ST_SHARDS
// cache lines).  Which slot computes a unit does not affect the unit's value, so the image is deterministic.
#include "zr_device.h"
#include "zr_launch.h"

namespace zr {

#ifndef ST_EXT_WAVES
#define ST_EXT_WAVES 4  /* waves per SIMD the EXTEND kernel's register budget must allow */
#endif
#ifndef ST_CHUNK
#define ST_CHUNK 256    /* rays a wave reserves per global atomic */
#endif
#ifndef ST_SHADE_WAVES
#define ST_SHADE_WAVES 2 /* 256-thread blocks per CU (= waves per SIMD) the SHADE kernel must fit */
#endif
#define ST_SHARDS 64     /* unit counters (ctl[16 + 32 * s]): a single contended word sustains only ~90 atomics/us */
#define ST_LDS_STACK 8
#define ST_OVERFLOW (ZR_STACK_DEPTH - ST_LDS_STACK)

enum { F_FIRST = 1u << 16, F_ACTIVE = 1u << 17 };  // meta.y: bounce | b_inner << 8 | flags

struct SEntry { uint32_t node; float tn; };

struct StreamBuf {
    double* ray;    // [6][P]
    double* hit_t;  // [P]
    uint2* hit_ki;  // [P] kind, index (kind = 0xFFFFFFFF: miss)
    double* beta;   // [3][P]
    double* L;      // [3][P]
    double* att0;   // [3][P]
    double* sum;    // [3][P]
    unsigned long long* key;  // [P]
    uint4* meta;    // [P] x = RNG draw index, y = bounce | b_inner << 8 | flags, z = work unit, w = sample
    const uint32_t* pixels;   // [n_pix] px | py << 16
    double* partial;          // [n_pix][lanes][3]
    unsigned int* ctl;        // [0] extend head, [1] active slots after the last SHADE, [2] iteration-cap hits, [16 + 32 s] unit counters
    uint32_t P, lanes, n_units, n_pix;
};

__device__ __forceinline__ double ldnt(const double* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(double* p, double v) { __builtin_nontemporal_store(v, p); }

// ---- begin a sample in a slot: camera ray + fresh path state --------------------------------------------
__device__ inline void begin_sample(const StreamBuf& B, const DCamera& cam, uint64_t seed, uint32_t slot, uint32_t unit, uint32_t sample,
                                    uint32_t& c_samp) {
    const uint32_t pk = B.pixels[unit / B.lanes];
    const int px = (int)(pk & 0xFFFFu), py = (int)(pk >> 16);
    Rng g; g.key = zr_stream_key(seed, (uint64_t)py * (uint64_t)cam.W + (uint64_t)px, (uint64_t)sample); g.k = 0; g.bounce = 0;
    Ray r = camera_ray(cam, px, py, g);
    const size_t P = B.P;
    stnt(B.ray + 0 * P + slot, r.o.x); stnt(B.ray + 1 * P + slot, r.o.y); stnt(B.ray + 2 * P + slot, r.o.z);
    stnt(B.ray + 3 * P + slot, r.d.x); stnt(B.ray + 4 * P + slot, r.d.y); stnt(B.ray + 5 * P + slot, r.d.z);
    B.key[slot] = g.key;
    uint4 m; m.x = (uint32_t)g.k; m.y = F_FIRST | F_ACTIVE; m.z = unit; m.w = sample;
    B.meta[slot] = m;
    c_samp++;
}

template <bool COUNT>
__global__ __launch_bounds__(256) void stream_init(StreamBuf B, DCamera cam, uint64_t seed, unsigned long long* __restrict__ gctr) {
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= B.P) return;
    const size_t P = B.P;
    for (int c = 0; c < 3; c++) B.sum[c * P + slot] = 0.0;
    uint32_t c_samp = 0;
    if (slot < B.n_units) begin_sample(B, cam, seed, slot, slot, slot % B.lanes, c_samp);
    else { uint4 m; m.x = 0; m.y = 0; m.z = 0; m.w = 0; B.meta[slot] = m; }
    if (COUNT && c_samp) atomicAdd(&gctr[0], (unsigned long long)c_samp);
}

// ---- EXTEND: closest hit for every active slot ------------------------------------------------------------
enum { X_IDLE = 0, X_NODE = 1, X_LEAF = 2, X_EXIT = 3 };

template <bool COUNT>
__global__ __launch_bounds__(64, ST_EXT_WAVES) void stream_extend(DScene sc, StreamBuf B, SEntry* __restrict__ overflow,
                                                                  unsigned long long* __restrict__ gctr) {
    __shared__ SEntry lstack[ST_LDS_STACK * 64];
    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    SEntry* gstack = overflow + (size_t)blockIdx.x * ST_OVERFLOW * 64 + lane;
    const double INF = __builtin_huge_val();
    const uint32_t NONE = 0xFFFFFFFFu;
    const size_t P = B.P;

    int st = X_IDLE;
    uint32_t slot = 0;
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
    double idx_ = 0, idy_ = 0, idz_ = 0, ox_ = 0, oy_ = 0, oz_ = 0;
    double tbest = INF;
    uint32_t kbest = NONE, ibest = 0, cur = NONE;
    int sp = 0;
    uint32_t pa_first = 0, pa_meta = 0, pb_first = 0, pb_meta = 0, pend_i = 0;
    Rng g; g.key = 0; g.k = 0; g.bounce = 0;  // only the medium test reads it
    bool work_left = true;
    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform: the private range of ray indices being handed out
    if (blockIdx.x == 0 && lane == 0) B.ctl[1] = 0;  // SHADE of this round recounts the active slots
    uint32_t c_nodes = 0, c_sph = 0, c_tri = 0, c_cube = 0, c_med = 0, c_seg = 0, c_hits = 0;
    unsigned long long s_exec[2] = {0, 0}, s_lanes[2] = {0, 0};

    auto pop_next = [&]() {
        cur = NONE;
        while (sp > 0) {
            sp--;
            SEntry e = sp < ST_LDS_STACK ? lstack[sp * 64 + lane] : gstack[(size_t)(sp - ST_LDS_STACK) * 64];
            if ((double)e.tn <= tbest) { cur = e.node; break; }
        }
    };
    auto finish = [&]() {  // traversal of this lane's ray is complete: publish the result
        stnt(B.hit_t + slot, tbest);
        uint2 ki; ki.x = kbest; ki.y = ibest;
        B.hit_ki[slot] = ki;
        if (COUNT && kbest != NONE) c_hits++;
        st = X_IDLE;
    };

    const unsigned long long iter_cap = (unsigned long long)P * 64ull + (1ull << 24);
    unsigned long long iter = 0;
    for (; iter < iter_cap; iter++) {
        const uint32_t lkind = (pa_meta >> 16) - 1u;
        const int n1 = __popcll(__ballot(st == X_NODE));
        const int n2t = __popcll(__ballot(st == X_LEAF && lkind == ZR_PRIM_TRIANGLE));
        const int n2s = __popcll(__ballot(st == X_LEAF && lkind == ZR_PRIM_SPHERE));
        const int n2g = __popcll(__ballot(st == X_LEAF)) - n2t - n2s;
        const int n0 = work_left ? __popcll(__ballot(st == X_IDLE)) : 0;
        const int n2 = n2t > n2s ? (n2t > n2g ? n2t : n2g) : (n2s > n2g ? n2s : n2g);
        if (n1 + n2 + n0 == 0) break;

        if (n0 >= 16 || (n0 > 0 && n0 >= n1 && n0 >= n2)) {
            // ================= FETCH: idle lanes take the next ray indices =================
            // rays are handed out from a wave-private chunk; one global atomic per ST_CHUNK rays (a single
            // contended word sustains only ~90 atomics/us on this chip)
            const unsigned long long idle = __ballot(st == X_IDLE);
            uint32_t n = (uint32_t)__popcll(idle);
            if (chunk_next >= chunk_end) {
                uint32_t nb = 0;
                if (lane == 0) nb = atomicAdd(&B.ctl[0], (unsigned int)ST_CHUNK);
                nb = __builtin_amdgcn_readfirstlane(nb);
                chunk_next = nb < B.P ? nb : B.P;
                chunk_end = nb + ST_CHUNK < B.P ? nb + ST_CHUNK : B.P;
                if (chunk_next >= chunk_end) work_left = false;
            }
            if (n > chunk_end - chunk_next) n = chunk_end - chunk_next;
            const uint32_t base = chunk_next;
            chunk_next += n;
            const uint32_t lim = base + n;
            if (st == X_IDLE) {
                const uint32_t my = base + (uint32_t)__popcll(idle & lt_mask);
                if (my < lim) {
                    const uint4 m = B.meta[my];
                    if (m.y & F_ACTIVE) {
                        slot = my;
                        ray.o = mk(ldnt(B.ray + 0 * P + my), ldnt(B.ray + 1 * P + my), ldnt(B.ray + 2 * P + my));
                        ray.d = mk(ldnt(B.ray + 3 * P + my), ldnt(B.ray + 4 * P + my), ldnt(B.ray + 5 * P + my));
                        g.key = B.key[my]; g.bounce = m.y & 0xFFu;
                        idx_ = 1.0 / ray.d.x; idy_ = 1.0 / ray.d.y; idz_ = 1.0 / ray.d.z;
                        ox_ = ray.o.x * idx_; oy_ = ray.o.y * idy_; oz_ = ray.o.z * idz_;
                        tbest = INF; kbest = NONE; cur = 0; sp = 0; pa_meta = 0; pb_meta = 0; pend_i = 0;
                        st = X_NODE;
                        if (COUNT) c_seg++;
                    }
                }
            }
        } else if (n1 >= n2) {
            // ================= NODE: one sibling-pair record per lane =================
            if (COUNT) { s_exec[0]++; s_lanes[0] += n1; }
            if (st == X_NODE) {
                const NodePair* np = sc.nodes + cur;
                const float4 q0 = reinterpret_cast<const float4*>(np)[0];
                const float4 q1 = reinterpret_cast<const float4*>(np)[1];
                const float4 q2 = reinterpret_cast<const float4*>(np)[2];
                const uint4 q3 = reinterpret_cast<const uint4*>(np)[3];
                if (COUNT) c_nodes += 2;
                double tn0, tf0, tn1, tf1;
                {
                    double a0 = fma((double)q0.x, idx_, -ox_), a1 = fma((double)q1.z, idx_, -ox_);
                    double b0 = fma((double)q0.y, idy_, -oy_), b1 = fma((double)q1.w, idy_, -oy_);
                    double c0 = fma((double)q0.z, idz_, -oz_), c1 = fma((double)q2.x, idz_, -oz_);
                    tn0 = fmax(fmax(fmin(a0, a1), fmin(b0, b1)), fmax(fmin(c0, c1), 0.001));
                    tf0 = fmin(fmin(fmax(a0, a1), fmax(b0, b1)), fmin(fmax(c0, c1), tbest));
                }
                {
                    double a0 = fma((double)q0.w, idx_, -ox_), a1 = fma((double)q2.y, idx_, -ox_);
                    double b0 = fma((double)q1.x, idy_, -oy_), b1 = fma((double)q2.z, idy_, -oy_);
                    double c0 = fma((double)q1.y, idz_, -oz_), c1 = fma((double)q2.w, idz_, -oz_);
                    tn1 = fmax(fmax(fmin(a0, a1), fmin(b0, b1)), fmax(fmin(c0, c1), 0.001));
                    tf1 = fmin(fmin(fmax(a0, a1), fmax(b0, b1)), fmin(fmax(c0, c1), tbest));
                }
                bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
                if (q3.z != 0 && (q3.z & 0xFFFFu) == 0) h0 = false;
                if (q3.w != 0 && (q3.w & 0xFFFFu) == 0) h1 = false;
                const bool swp = h0 && h1 && tn1 < tn0;
                const uint32_t cA = swp ? q3.y : q3.x, cB = swp ? q3.x : q3.y;
                const uint32_t mA = swp ? q3.w : q3.z, mB = swp ? q3.z : q3.w;
                const bool hA = swp ? h1 : h0, hB = swp ? h0 : h1;
                const double tnB = swp ? tn0 : tn1;
                uint32_t next = NONE;
                pa_meta = 0; pb_meta = 0; pend_i = 0;
                if (hA) {
                    if (mA != 0) { pa_first = cA; pa_meta = mA; } else next = cA;
                }
                if (hB) {
                    if (mB != 0) {
                        if (pa_meta == 0) { pa_first = cB; pa_meta = mB; } else { pb_first = cB; pb_meta = mB; }
                    } else if (next == NONE) {
                        next = cB;
                    } else {
                        SEntry e; e.node = cB; e.tn = __double2float_rd(tnB);
                        if (sp < ST_LDS_STACK) lstack[sp * 64 + lane] = e; else gstack[(size_t)(sp - ST_LDS_STACK) * 64] = e;
                        sp++;
                    }
                }
                cur = next;
                if (pa_meta != 0) st = X_LEAF;
                else {
                    if (cur == NONE) pop_next();
                    if (cur == NONE) finish();
                }
            }
        } else {
            // ================= LEAF: one primitive per lane, the kind with most waiting lanes =================
            if (COUNT) { s_exec[1]++; s_lanes[1] += n2; }
            const bool is_leaf = st == X_LEAF;
            const bool do_tri = n2t == n2;
            const bool do_sph = !do_tri && n2s == n2;
            bool tested = false;
            if (do_tri) {
                if (is_leaf && lkind == ZR_PRIM_TRIANGLE) {
                    double t;
                    if (COUNT) c_tri++;
                    if (triangle_t(sc.tri_v + (size_t)(pa_first + pend_i) * 9, ray, 0.001, tbest, t)) { tbest = t; kbest = lkind; ibest = pa_first + pend_i; }
                    tested = true;
                }
            } else if (do_sph) {
                if (is_leaf && lkind == ZR_PRIM_SPHERE) {
                    double t;
                    if (COUNT) c_sph++;
                    if (sphere_t(sc.spheres + (size_t)(pa_first + pend_i) * 4, ray, 0.001, tbest, t)) { tbest = t; kbest = lkind; ibest = pa_first + pend_i; }
                    tested = true;
                }
            } else if (is_leaf && lkind != ZR_PRIM_TRIANGLE && lkind != ZR_PRIM_SPHERE) {
                double t;
                if (COUNT) {
                    uint32_t kk = lkind;
                    if (kk == ZR_KIND_WRAPPED) kk = sc.wrapped[pa_first + pend_i].type;
                    if (kk == ZR_PRIM_SPHERE) c_sph++; else if (kk == ZR_PRIM_TRIANGLE) c_tri++; else if (kk == ZR_PRIM_CUBE) c_cube++; else c_med++;
                }
                if (object_t(sc, lkind, pa_first + pend_i, ray, 0.001, tbest, g, t)) { tbest = t; kbest = lkind; ibest = pa_first + pend_i; }
                tested = true;
            }
            if (tested) {
                pend_i++;
                if (pend_i >= (pa_meta & 0xFFFFu)) {
                    pa_first = pb_first; pa_meta = pb_meta; pb_meta = 0; pend_i = 0;
                    if (pa_meta == 0) {
                        if (cur == NONE) pop_next();
                        if (cur != NONE) st = X_NODE; else finish();
                    }
                }
            }
        }
    }
    if (iter >= iter_cap && lane == 0) atomicAdd(&B.ctl[2], 1u);
    if (COUNT) {
        atomicAdd(&gctr[1], (unsigned long long)c_seg);
        atomicAdd(&gctr[2], (unsigned long long)c_nodes);
        atomicAdd(&gctr[3], (unsigned long long)c_sph);
        atomicAdd(&gctr[4], (unsigned long long)c_tri);
        atomicAdd(&gctr[5], (unsigned long long)c_cube);
        atomicAdd(&gctr[6], (unsigned long long)c_med);
        atomicAdd(&gctr[7], (unsigned long long)c_hits);
        if (lane == 0) for (int k = 0; k < 2; k++) { atomicAdd(&gctr[9 + 2 * k], s_exec[k]); atomicAdd(&gctr[10 + 2 * k], s_lanes[k]); }
    }
}

// ---- SHADE: one segment of every active slot ------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256, ST_SHADE_WAVES) void stream_shade(DScene sc, DCamera cam, DEnv env, uint64_t seed, StreamBuf B,
                                                    unsigned long long* __restrict__ gctr) {
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    if (slot == 0) B.ctl[0] = 0;  // EXTEND of the next round starts from ray 0 again
    bool active_after = false, want_unit = false;
    uint32_t c_samp = 0; unsigned long long c_draws = 0;
    if (slot < B.P) {
        uint4 m = B.meta[slot];
        if (m.y & F_ACTIVE) {
            const size_t P = B.P;
            const uint32_t NONE = 0xFFFFFFFFu;
            Ray ray;
            ray.o = mk(ldnt(B.ray + 0 * P + slot), ldnt(B.ray + 1 * P + slot), ldnt(B.ray + 2 * P + slot));
            ray.d = mk(ldnt(B.ray + 3 * P + slot), ldnt(B.ray + 4 * P + slot), ldnt(B.ray + 5 * P + slot));
            const double t = ldnt(B.hit_t + slot);
            const uint2 ki = B.hit_ki[slot];
            Rng g; g.key = B.key[slot]; g.k = m.x; g.bounce = (m.y & 0xFFu) + 1u;  // this closest-hit query is complete
            int b_inner = (int)((m.y >> 8) & 0xFFu);
            const bool first = (m.y & F_FIRST) != 0;
            const int depth_inner = cam.max_depth - 1;
            V3 L = mk(0, 0, 0), beta = mk(1, 1, 1), att0 = mk(1, 1, 1);
            if (!first) {
                L = mk(B.L[slot], B.L[P + slot], B.L[2 * P + slot]);
                beta = mk(B.beta[slot], B.beta[P + slot], B.beta[2 * P + slot]);
                att0 = mk(B.att0[slot], B.att0[P + slot], B.att0[2 * P + slot]);
            }
            V3 sum = mk(B.sum[slot], B.sum[P + slot], B.sum[2 * P + slot]);
            bool ended = false, cont_first = false;
            V3 contrib = mk(0, 0, 0);
            Ray nr; nr.o = mk(0, 0, 0); nr.d = mk(0, 0, 1);
            if (ki.x == NONE) {
                V3 bg = background(sc, env, ray.d);
                contrib = first ? bg : att0 * (L + beta * bg);   // camera.hpp:520 / 941,1000
                ended = true;
            } else {
                Rec rec;
                object_rec(sc, ki.x, ki.y, ray, t, rec);
                V3 em = emitted(sc, rec);
                V3 att;
                const bool sc_ok = scatter(sc, ray, rec, att, nr, g);
                if (first) {  // ray_color_from_hit, camera.hpp:989-1004
                    sum = sum + em;
                    if (!sc_ok || depth_inner <= 0) ended = true;
                    else { att0 = att; L = mk(0, 0, 0); beta = mk(1, 1, 1); b_inner = 0; cont_first = true; }
                } else {      // body of ray_color's loop, camera.hpp:944-983
                    L = L + beta * em;
                    bool stop = !sc_ok;
                    if (!stop) {
                        beta = beta * att;
                        if (b_inner > 10) {
                            if (len(beta) < 0.0001) stop = true;
                            else {
                                double p = fmax(fmax(beta.x, beta.y), beta.z);
                                p = clampd(p, 0.05, 0.95);
                                if (g.next() > p) stop = true; else beta = beta * (1 / p);
                            }
                        }
                    }
                    if (!stop) { b_inner++; if (b_inner >= depth_inner) stop = true; }
                    if (stop) { contrib = att0 * L; ended = true; }
                }
            }
            if (!ended) {
                // the path continues: publish the scattered ray and the path state
                stnt(B.ray + 0 * P + slot, nr.o.x); stnt(B.ray + 1 * P + slot, nr.o.y); stnt(B.ray + 2 * P + slot, nr.o.z);
                stnt(B.ray + 3 * P + slot, nr.d.x); stnt(B.ray + 4 * P + slot, nr.d.y); stnt(B.ray + 5 * P + slot, nr.d.z);
                B.L[slot] = L.x; B.L[P + slot] = L.y; B.L[2 * P + slot] = L.z;
                B.beta[slot] = beta.x; B.beta[P + slot] = beta.y; B.beta[2 * P + slot] = beta.z;
                if (cont_first) { B.att0[slot] = att0.x; B.att0[P + slot] = att0.y; B.att0[2 * P + slot] = att0.z; }
                if (first) { B.sum[slot] = sum.x; B.sum[P + slot] = sum.y; B.sum[2 * P + slot] = sum.z; }
                m.x = (uint32_t)g.k; m.y = (g.bounce & 0xFFu) | ((uint32_t)b_inner << 8) | F_ACTIVE;
                B.meta[slot] = m;
                active_after = true;
            } else {
                sum = sum + contrib;
                if (COUNT) c_draws += g.k;
                // next sample of this unit, or a new unit
                uint32_t sample = m.w + B.lanes;
                if (sample < (uint32_t)cam.spp) {
                    B.sum[slot] = sum.x; B.sum[P + slot] = sum.y; B.sum[2 * P + slot] = sum.z;
                    begin_sample(B, cam, seed, slot, m.z, sample, c_samp);
                    active_after = true;
                } else {
                    double* pp = B.partial + (size_t)m.z * 3;  // = [pixel][lane][3]
                    pp[0] = sum.x; pp[1] = sum.y; pp[2] = sum.z;
                    B.sum[slot] = 0.0; B.sum[P + slot] = 0.0; B.sum[2 * P + slot] = 0.0;
                    want_unit = true;
                }
            }
        }
    }
    // hand out new work units: one atomic per wave on this block's shard
    {
        const unsigned long long wm = __ballot(want_unit);
        if (wm != 0ull) {
            const uint32_t shard = blockIdx.x % ST_SHARDS;
            const int wl = threadIdx.x & 63;
            uint32_t k0 = 0;
            if (wl == (int)__builtin_ctzll(wm)) k0 = atomicAdd(&B.ctl[16 + 32 * shard], (unsigned int)__popcll(wm));
            k0 = __shfl(k0, (int)__builtin_ctzll(wm), 64);
            if (want_unit) {
                const unsigned long long k = (unsigned long long)k0 + (unsigned long long)__popcll(wm & ((1ull << wl) - 1ull));
                const unsigned long long u = (unsigned long long)B.P + k * ST_SHARDS + shard;
                if (u < (unsigned long long)B.n_units) { begin_sample(B, cam, seed, slot, (uint32_t)u, (uint32_t)(u % B.lanes), c_samp); active_after = true; }
                else { uint4 m0; m0.x = 0; m0.y = 0; m0.z = 0; m0.w = 0; B.meta[slot] = m0; }
            }
        }
    }
    const unsigned long long am = __ballot(active_after);
    if ((threadIdx.x & 63) == 0 && am != 0ull) atomicAdd(&B.ctl[1], (unsigned int)__popcll(am));
    if (COUNT) {
        if (c_samp) atomicAdd(&gctr[0], (unsigned long long)c_samp);
        if (c_draws) atomicAdd(&gctr[8], c_draws);
    }
}

__global__ __launch_bounds__(256) void stream_reduce(StreamBuf B, DCamera cam, double* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B.n_pix) return;
    const uint32_t pk = B.pixels[i];
    const int px = (int)(pk & 0xFFFFu), py = (int)(pk >> 16);
    const double* pp = B.partial + (size_t)i * B.lanes * 3;
    double sx = 0, sy = 0, sz = 0;
    for (uint32_t j = 0; j < B.lanes; j++) { sx += pp[j * 3]; sy += pp[j * 3 + 1]; sz += pp[j * 3 + 2]; }
    const double scale = 1.0 / cam.spp;  // camera.hpp:437,531
    double* o = out + ((size_t)py * cam.W + px) * 3;
    o[0] = sx * scale; o[1] = sy * scale; o[2] = sz * scale;
}

// ---- host-side launch helpers -----------------------------------------------------------------------------------
size_t stream_ctl_words() { return 16 + 32 * ST_SHARDS; }
size_t stream_overflow_bytes(int blocks) { return (size_t)blocks * ST_OVERFLOW * 64 * sizeof(SEntry); }

int stream_extend_blocks() {
    int dev = 0, cus = 256, per_cu = 16;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stream_extend<false>, 64, 0) != hipSuccess || per_cu < 1) per_cu = 16;
    return cus * per_cu;
}

// layout of the slot pool inside one allocation; returns bytes needed
size_t stream_pool_bytes(uint32_t P) {
    return (size_t)P * (6 * 8 + 8 + 8 + 4 * 3 * 8 + 8 + 16) + 4096;
}

static StreamBuf make_buf(void* pool, uint32_t P, uint32_t lanes, uint32_t n_units, uint32_t n_pix, const uint32_t* pixels, double* partial,
                          unsigned int* ctl) {
    StreamBuf B;
    unsigned char* p = (unsigned char*)pool;
    B.ray = (double*)p; p += (size_t)P * 48;
    B.hit_t = (double*)p; p += (size_t)P * 8;
    B.hit_ki = (uint2*)p; p += (size_t)P * 8;
    B.beta = (double*)p; p += (size_t)P * 24;
    B.L = (double*)p; p += (size_t)P * 24;
    B.att0 = (double*)p; p += (size_t)P * 24;
    B.sum = (double*)p; p += (size_t)P * 24;
    B.key = (unsigned long long*)p; p += (size_t)P * 8;
    B.meta = (uint4*)p;
    B.pixels = pixels; B.partial = partial; B.ctl = ctl;
    B.P = P; B.lanes = lanes; B.n_units = n_units; B.n_pix = n_pix;
    return B;
}

hipError_t stream_render(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, void* pool, uint32_t P, uint32_t lanes,
                         uint32_t n_pix, const uint32_t* d_pixels, double* d_partial, unsigned int* d_ctl, void* d_overflow, int extend_blocks,
                         double* out, unsigned long long* gctr, bool count, hipStream_t stream, StreamTimer* timer,
                         unsigned int* h_active, volatile const uint8_t* keep_going, int* rounds_out) {
    const uint32_t n_units = n_pix * lanes;
    StreamBuf B = make_buf(pool, P, lanes, n_units, n_pix, d_pixels, d_partial, d_ctl);
    hipError_t e;
    if ((e = hipMemsetAsync(d_ctl, 0, stream_ctl_words() * sizeof(unsigned int), stream)) != hipSuccess) return e;
    const unsigned pblocks = (P + 255) / 256;
    if (timer) timer->begin(stream, 0);
    if (count) hipLaunchKernelGGL(stream_init<true>, dim3(pblocks), dim3(256), 0, stream, B, cam, seed, gctr);
    else hipLaunchKernelGGL(stream_init<false>, dim3(pblocks), dim3(256), 0, stream, B, cam, seed, gctr);
    if (timer) timer->end(stream, 0);
    *h_active = 1;
    int rounds = 0;
    const int check_every = 8;
    const int eb = (int)((P + 63) / 64 < (uint32_t)extend_blocks ? (P + 63) / 64 : (uint32_t)extend_blocks);
    bool cancelled = false;
    for (;;) {
        for (int k = 0; k < check_every; k++) {
            if (timer) timer->begin(stream, 1);
            if (count) hipLaunchKernelGGL(stream_extend<true>, dim3(eb), dim3(64), 0, stream, sc, B, (SEntry*)d_overflow, gctr);
            else hipLaunchKernelGGL(stream_extend<false>, dim3(eb), dim3(64), 0, stream, sc, B, (SEntry*)d_overflow, gctr);
            if (timer) timer->end(stream, 1);
            if (timer) timer->begin(stream, 2);
            if (count) hipLaunchKernelGGL(stream_shade<true>, dim3(pblocks), dim3(256), 0, stream, sc, cam, env, seed, B, gctr);
            else hipLaunchKernelGGL(stream_shade<false>, dim3(pblocks), dim3(256), 0, stream, sc, cam, env, seed, B, gctr);
            if (timer) timer->end(stream, 2);
            rounds++;
        }
        if ((e = hipMemcpyAsync(h_active, d_ctl + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, stream)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) break;
        if (*h_active == 0) break;
        if (keep_going && *keep_going == 0) { cancelled = true; break; }
        if (rounds > (1 << 22)) { e = hipErrorLaunchFailure; break; }
    }
    if (e != hipSuccess) return e;
    if (timer) timer->begin(stream, 3);
    hipLaunchKernelGGL(stream_reduce, dim3((n_pix + 255) / 256), dim3(256), 0, stream, B, cam, out);
    if (timer) timer->end(stream, 3);
    if (rounds_out) *rounds_out = cancelled ? -rounds : rounds;
    return hipGetLastError();
}

}  // namespace zr
