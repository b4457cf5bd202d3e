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
//   end:    REDUCE  every finished sample wrote its radiance to samples[pixel][s]; one fixed-order sum per pixel
//                   gives the mean — no atomics on radiance anywhere, the image is bit-reproducible
//
// Work units: unit u = one primary sample = (pixel u / spp of the frame's pixel list, sample u % spp).  Slot k starts on
// unit k (64 consecutive slots = one pixel: coherent primary rays); a slot whose path ends writes the sample's
// radiance to samples[u] and takes the next unit from one of ST_SHARDS interleaved counters (one atomic per wave per
// round, spread over ST_SHARDS cache lines).  Every sample is written exactly once, by whichever slot computed it, and
// REDUCE sums a pixel's samples in an order that depends only on spp — so the image is bit-reproducible and does not
// depend on the pool size, on scheduling, or on how the frame is sharded over GPUs.
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
#define ST_SHADE_WAVES 4 /* waves per SIMD the SHADE kernel must fit (128 VGPRs): its natural allocation sits at 126-129 */
#endif
#ifndef ST_EXT_WAVES_LEAN
#define ST_EXT_WAVES_LEAN 6  /* same for the triangles-and-spheres-only build of EXTEND */
#endif
#ifndef ST_EXT_WAVES_MID
#define ST_EXT_WAVES_MID 4   /* ... and for the build that adds bare / placed cubes and unwrapped media (cfg5): at 5 waves (96 VGPRs) it spills 18 registers and is 1.5 % slower, at 6 waves 14 % slower */
#endif
#ifndef ST_FETCH_MIN
#define ST_FETCH_MIN 16  /* idle lanes that trigger a refill even when another phase has more ready lanes */
#endif
#ifndef ST_LEAF_ONE_KIND
#define ST_LEAF_ALL 1   /* a LEAF iteration serves EVERY leaf kind that has waiting lanes, kind by kind under wave-uniform guards (round 3), instead of only
                           the kind with most lanes: the few lanes at the other kind (cfg3: the ground sphere every ray meets) no longer sit out NODE
                           iterations waiting for company.  cfg3 EXTEND 213.1 -> 204.6 ms per frame with the 1:1 bias below (2:3: 206.2; the old
                           one-kind rule with its 1:2 bias: 213.1), cfg2 46.7 -> 46.2; image bit-identical (profiles/r3_experiments_ab.txt) */
#endif
#ifndef ST_BIAS_NODE
#define ST_BIAS_NODE 1  /* NODE runs when ready NODE lanes x ST_BIAS_NODE >= ready LEAF lanes x ST_BIAS_LEAF (LEAF lanes = all kinds together with ST_LEAF_ALL) */
#ifdef ST_LEAF_ALL
#define ST_BIAS_LEAF 1
#else
#define ST_BIAS_LEAF 2  /* one kind per LEAF iteration: favouring LEAF 2:1 was 3 % faster on cfg3 than 1:1 (a tested leaf shrinks tbest and culls the stack) */
#endif
#endif
#define ST_SHARDS 64     /* unit counters (ctl[16 + 32 * s]): a single contended word sustains only ~90 atomics/us */
#ifndef ST_LDS_STACK
#define ST_LDS_STACK 12
#endif

enum { F_FIRST = 1u << 16, F_ACTIVE = 1u << 17, F_L0 = 1u << 18,
       F_LZERO = 1u << 19 /* SF_L not written yet: it is (0,0,0) */, F_BONE = 1u << 20 /* SF_BETA not written yet: it is (1,1,1) */ };  // meta.y: bounce | b_inner << 8 | flags (F_L0: SF_SUM holds the primary hit's emission)

struct SEntry { uint32_t node; float tn; };

// Slot pool layout: array of 64-slot blocks, each block = SF_N rows of 64 x 8 bytes ([block][field][lane]).  A wave
// that works on 64 consecutive slots touches one contiguous 12 KB block, every field access is one coalesced
// 512-B row, and all of a slot's state shares a page (a plain [field][P] layout put each of the 23 fields 128 MB
// apart and ran the SHADE stage at 1.7 TB/s).
enum { SF_RAY = 0, SF_HIT_T = 6, SF_HIT_KI = 7, SF_BETA = 8, SF_L = 11, SF_ATT0 = 14, SF_SUM = 17, SF_KEY = 20,
       SF_MA = 21 /* RNG draw index, bounce | b_inner << 8 | flags */, SF_MB = 22 /* work unit */, SF_N = 24 };

struct StreamBuf {
    double* pool;             // [P / 64][SF_N][64] 8-byte cells
    const uint32_t* pixels;   // [n_pix] px | py << 16
    double* samples;          // [n_pix][spp][3] radiance of every primary sample
    unsigned int* ctl;        // [2] iteration-cap hits; per shard s: [16 + 32 s] unit counter, [16 + 32 s + 8] EXTEND chunk head,
                              // [16 + 32 s + 16] active slots after the last SHADE (ST_SHARDS words on separate cache lines:
                              // a single contended word sustains only ~90 atomics/us)
    unsigned int* uctl;       // work-unit counters [32 s], shared by the two half pools
    uint32_t P, spp, n_units, n_pix;
    uint32_t unit0;           // slot k of this pool starts on unit unit0 + k
    uint32_t unit_base;       // first dynamically assigned unit (= slots of both pools)
    uint32_t unit_chunk;      // 0: units are striped over the shards (unit_base + k * ST_SHARDS + shard); G > 0: XCD-affine hand-out, see st_unit_of
    uint32_t shard_k0;        // affine hand-out: index in its shard's sequence of the first unit a shard deals dynamically (= slots per shard)
    uint2* kend;              // reflection / refraction split: per unit (draws, segments) the beauty path consumed
    unsigned char* cls;       // reflection / refraction split: per unit 1 = reflection, 2 = refraction, 0 = no contribution
    unsigned long long* cpart; // reflection / refraction split: per SHADE block (samples, segments, hits, draws), owned by that block
    __device__ __forceinline__ double* cell(int f, uint32_t slot) const { return pool + ((size_t)(slot >> 6) * SF_N + f) * 64 + (slot & 63u); }
    __device__ __forceinline__ double ld(int f, uint32_t slot) const { return *cell(f, slot); }
    __device__ __forceinline__ void st(int f, uint32_t slot, double v) const { *cell(f, slot) = v; }
    __device__ __forceinline__ uint2 ld2(int f, uint32_t slot) const { return *reinterpret_cast<const uint2*>(cell(f, slot)); }
    __device__ __forceinline__ void st2(int f, uint32_t slot, uint2 v) const { *reinterpret_cast<uint2*>(cell(f, slot)) = v; }
    __device__ __forceinline__ V3 ld3(int f, uint32_t slot) const { return mk(ld(f, slot), ld(f + 1, slot), ld(f + 2, slot)); }
    __device__ __forceinline__ void st3(int f, uint32_t slot, V3 v) const { st(f, slot, v.x); st(f + 1, slot, v.y); st(f + 2, slot, v.z); }
};

// A slot addressed RELATIVE TO ITS 256-SLOT BLOCK (round 4).  SHADE and INIT work on the slots of one block per workgroup, so `base` is the same in every
// thread (a scalar register pair), `off` a 32-bit byte offset below 48 KB and the field a constant: every row access becomes
// `global_load/store v, v_off, s[base] offset:imm` with ONE vector register of address for all 23 rows.  Through StreamBuf::cell() every row had a 64-bit
// address of its own, and the rows a segment reads and writes back kept theirs alive across the whole kernel (12+ registers, part of them spilled in the
// lean build): SHADE's speed follows its occupancy (profiles/r4_experiments_ab.txt), so registers are time.
struct SlotAt {
    const char* base; uint32_t off;
    __device__ __forceinline__ const char* at(int f) const { return base + (off + (uint32_t)f * 512u); }
    __device__ __forceinline__ double ld(int f) const { return *reinterpret_cast<const double*>(at(f)); }
    __device__ __forceinline__ void st(int f, double v) const { *reinterpret_cast<double*>(const_cast<char*>(at(f))) = v; }
    __device__ __forceinline__ uint2 ld2(int f) const { return *reinterpret_cast<const uint2*>(at(f)); }
    __device__ __forceinline__ void st2(int f, uint2 v) const { *reinterpret_cast<uint2*>(const_cast<char*>(at(f))) = v; }
    __device__ __forceinline__ V3 ld3(int f) const { return mk(ld(f), ld(f + 1), ld(f + 2)); }
    __device__ __forceinline__ void st3(int f, V3 v) const { st(f, v.x); st(f + 1, v.y); st(f + 2, v.z); }
};
// slot `block * 256 + src` (src < 256); `block` must be wave-uniform
__device__ __forceinline__ SlotAt slot_at(const StreamBuf& B, uint32_t block, uint32_t src) {
    SlotAt a;
    a.base = reinterpret_cast<const char*>(B.pool) + (size_t)block * (size_t)(4 * SF_N * 512);
    a.off = (src >> 6) * (uint32_t)(SF_N * 512) + (src & 63u) * 8u;
    return a;
}


// Which unit is the k-th of shard `shard`.
//  striped (G = 0): unit_base + k * ST_SHARDS + shard — every shard sweeps the whole frame, one unit in ST_SHARDS.
//  affine  (G > 0): the frame's units (pixel list in tile order x spp) are cut into chunks of G units and chunk c belongs to
//    the shard at position c % ST_SHARDS of a round; within its chunks a shard deals consecutive units.  A shard is served by
//    the SHADE blocks and the EXTEND waves with blockIdx % ST_SHARDS == shard, i.e. by ONE XCD class (blocks b and b + 8 share
//    an XCD: MI355X_MICROARCH.md, workgroup dispatch), and position (shard % 8) * 8 + shard / 8 puts the eight shards of a class
//    side by side: at any time an XCD's L2 sees the rays of ~8 neighbouring chunks (screen tiles) instead of every 64th unit of
//    a front that spans the frame.  The image cannot depend on it: samples[unit] is written once, by whichever slot got the unit.
__host__ __device__ inline unsigned long long st_unit_of(uint32_t G, uint32_t unit_base, uint32_t shard, unsigned long long k) {
    if (G == 0) return (unsigned long long)unit_base + k * ST_SHARDS + shard;
    const uint32_t k32 = (uint32_t)k, q = k32 / G, r = k32 - q * G;
    const uint32_t pos = (shard & 7u) * (ST_SHARDS / 8u) + (shard >> 3);
    return ((unsigned long long)q * ST_SHARDS + pos) * G + r;
}
// the unit slot g (index over all sub-pools) starts on
__host__ __device__ inline unsigned long long st_first_unit(uint32_t G, uint32_t g) {
    if (G == 0) return g;
    const uint32_t c = g >> 8;
    return st_unit_of(G, 0, c % ST_SHARDS, (unsigned long long)(c / ST_SHARDS) * 256u + (g & 255u));
}

// ---- begin a sample in a slot: camera ray + fresh path state --------------------------------------------
__device__ inline void begin_sample(const StreamBuf& B, const DCamera& cam, uint64_t seed, const SlotAt& S, uint32_t unit, uint32_t& c_samp) {
    const uint32_t pix_i = unit / B.spp, sample = unit - pix_i * B.spp;
    const uint32_t pk = B.pixels[pix_i];
    const int px = (int)(pk & 0xFFFFu), py = (int)(pk >> 16);
    Rng g; g.key = zr_stream_key(seed, (uint64_t)py * (uint64_t)cam.W + (uint64_t)px, (uint64_t)sample); g.k = 0; g.bounce = 0;
    Ray r = camera_ray(cam, px, py, g);
    S.st3(SF_RAY, r.o); S.st3(SF_RAY + 3, r.d);
    S.st(SF_KEY, __longlong_as_double((long long)g.key));
    uint2 ma; ma.x = (uint32_t)g.k; ma.y = F_FIRST | F_ACTIVE;
    uint2 mb; mb.x = unit; mb.y = 0;
    S.st2(SF_MA, ma); S.st2(SF_MB, mb);
    c_samp++;
}

template <bool COUNT>
__global__ __launch_bounds__(256) void stream_init(StreamBuf B, DCamera cam, uint64_t seed, unsigned long long* __restrict__ gctr) {
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= B.P) return;
    uint32_t c_samp = 0;
    const unsigned long long u0 = st_first_unit(B.unit_chunk, B.unit0 + slot);
    const SlotAt S = slot_at(B, blockIdx.x, threadIdx.x);
    if (u0 < (unsigned long long)B.n_units) begin_sample(B, cam, seed, S, (uint32_t)u0, c_samp);
    else { uint2 z; z.x = 0; z.y = 0; S.st2(SF_MA, z); }
    if (COUNT && c_samp) atomicAdd(&gctr[0], (unsigned long long)c_samp);
}

// ---- EXTEND: closest hit for every active slot ------------------------------------------------------------
// Walks the 4-wide tree (zr_device_types.h: FP32 root in the kernel arguments, 64-byte nodes with boxes on an 8-bit
// grid below it).  The kernel sits where VALU issue and the node / primitive fetches are in balance (rocprofv3:
// SQ_INSTS_VALU x 4 clocks / 1024 SIMDs ~ its duration — and yet 6.8 % fewer vector instructions per launch, round 3's
// node step, left the duration where it was, while 4 / 9 / 17 % MORE cost 2 / 10 / 15 %: profiles/r3_experiments_ab.txt;
// 128-byte FP32 nodes with fewer operations per visit were 15 % slower), so the slab test is arranged for few operations
// AND few bytes, and it is CONSERVATIVE: a superset of box hits cannot change the closest
// primitive, and every primitive test stays FP64.
//   per ray:   id = 1 / (float)d (float, taken as exact: it perturbs t by a relative 2^-23),  c = (float)(-o * id)
//   per plane: t = fmaf(P, id, c)                          (root: P is an FP32 plane)
//              t = fmaf(q, a, b), a = scale * id (exact, scale is a power of two), b = fmaf(origin, id, c)
//   error:     |t - (P - o) id| <= 2^-24 (|t| + |b| + |c|), and |b| <= |t| + 255 |a|
// The absolute part of that bound is folded, per axis, into the constants: the plane a ray ENTERS an axis' slab through
// (the lower one when id > 0) uses c_n = c - 2^-22 |c|, the one it leaves through c_f = c + 2^-22 |c| (and b_n, b_f move by a
// further 2^-15 |a|); a slack shared by the three axes would let an axis with a tiny d inflate the other two.  The
// relative part lowers the entry distance and raises the exit distance by |t| 2^-20.  An axis whose 1/d or o/d
// leaves the float range gets id = 0, c = NaN: its planes evaluate to NaN, which fminf/fmaxf ignore, i.e. the slab
// is dropped (conservative).
enum { X_IDLE = 0, X_NODE = 1, X_LEAF = 2 };
#define X_LEAF_BIT ZR_REF_LEAF

__device__ __forceinline__ void cswap(float& ta, uint32_t& ra, float& tb, uint32_t& rb) {
    const bool sw = tb < ta;
    const float t0 = sw ? tb : ta, t1 = sw ? ta : tb;
    const uint32_t r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ta = t0; tb = t1; ra = r0; rb = r1;
}

// LEVEL: which leaf kinds the build knows, chosen per scene at commit (zr_scene::leaf_level) so that a world pays only for
// the code — and the registers — of what it contains:
//   0  bare triangles and spheres (cfg2, cfg3)
//   1  + bare cubes, placed cubes (cube -> [rotate_y] -> translate) and media in an unwrapped sphere or cube (cfg5)
//   2  + objects under arbitrary wrapper chains and wrapped media (the op-list interpreter)
//   3  + placed runs of triangles (two-level BVH: zr_device.h instance_t; its private stack costs this build registers)
// Stack: ST_LDS_STACK entries per lane in LDS, deeper ones in this wave's slab of `overflow` (ovf_levels x 64 entries); the
// host sizes the slab from the exact worst-case demand of the committed tree (Flattener::stack_demand), so no push can
// leave it.
// ST_EXT_GROUP waves form a workgroup; they share nothing (own stack slice, ray chunks and state machine).  1 is the measured
// best: groups of 8 sharing an LDS copy of the tree's first three levels (84 nodes) were tried in round 2 — the copy's reads
// and the extra branch in the node step cost more than the requests they saved (cfg3 3.31 -> 3.60 ms per launch, cfg5 -3 %
// against +1 % for the grouping alone): profiles/r2_experiments_ab.txt.
#ifndef ST_EXT_GROUP
#define ST_EXT_GROUP 1
#endif
template <bool COUNT, int LEVEL>
__global__ __launch_bounds__(64 * ST_EXT_GROUP, LEVEL >= 2 ? ST_EXT_WAVES : (LEVEL == 1 ? ST_EXT_WAVES_MID : ST_EXT_WAVES_LEAN)) void stream_extend(DScene sc, StreamBuf B, SEntry* __restrict__ overflow,
                                                                  uint32_t ovf_levels, unsigned long long* __restrict__ gctr) {
    __shared__ SEntry lstack[ST_EXT_GROUP * ST_LDS_STACK * 64];
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = blockIdx.x * ST_EXT_GROUP + (threadIdx.x >> 6);
    const int lbase = (int)(threadIdx.x >> 6) * ST_LDS_STACK * 64 + lane;   // this lane's column of this wave's stack slice
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    SEntry* gstack = overflow + (size_t)wave_id * ovf_levels * 64 + lane;
    const double INF = __builtin_huge_val();
    const float INFf = __builtin_huge_valf();
    const uint32_t NONE = 0xFFFFFFFFu;

    int st = X_IDLE;
    uint32_t slot = 0;
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
    float idx_ = 0, idy_ = 0, idz_ = 0;
    float cnx = 0, cny = 0, cnz = 0, cfx = 0, cfy = 0, cfz = 0;  // per axis: c for the plane the ray ENTERS through and for the one it LEAVES through (slack folded in)
    double tbest = INF;
    float tbest_f = INFf;
    uint32_t kbest = NONE, ibest = 0;
    uint32_t inst_cur = 0;  // level 3: 0 = the lane walks the world's tree with the world ray; p + 1 = it is inside placement p, with the ray mapped into the run's space
    uint32_t cur = 0;       // X_NODE: quad index; X_LEAF: leaf reference (kind << 28 | (count - 1) << 24 | first)
    uint32_t pend_i = 0;
    int sp = 0;
    Rng g; g.key = 0; g.k = 0; g.bounce = 0;  // only the medium test reads it
    bool work_left = true;
    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform: the private range of ray indices being handed out
    uint32_t head_shard = wave_id % ST_SHARDS, shards_tried = 0;
    if (wave_id == 0 && lane < ST_SHARDS) B.ctl[16 + 32 * lane + 16] = 0;  // SHADE of this round recounts the active slots
    uint32_t c_nodes = 0, c_sph = 0, c_tri = 0, c_cube = 0, c_med = 0, c_seg = 0, c_hits = 0;
    unsigned long long s_exec[2] = {0, 0}, s_lanes[2] = {0, 0};

    auto finish = [&]() {  // traversal of this lane's ray is complete: publish the result
        B.st(SF_HIT_T, slot, tbest);
        uint2 ki; ki.x = kbest; ki.y = ibest;
        B.st2(SF_HIT_KI, slot, ki);
        if (COUNT && kbest != NONE) c_hits++;
        st = X_IDLE;
    };
// stack helpers are macros, not lambdas: a lambda capturing the __shared__ array by reference turns its accesses into
// flat-pointer accesses (and trips an LLVM verifier error on gfx950)
#define ZR_LDS_LEVELS ST_LDS_STACK
#define ZR_PUSH(REF, TN)                                                                                        \
    {                                                                                                           \
        SEntry e_; e_.node = (REF); e_.tn = (TN);                                                               \
        if (sp < ZR_LDS_LEVELS) lstack[sp * 64 + lbase] = e_; else gstack[(size_t)(sp - ZR_LDS_LEVELS) * 64] = e_; \
        sp++;                                                                                                   \
    }
// nearest deferred entry that can still matter, else the ray is done
#define ZR_POP_NEXT()                                                                                           \
    for (;;) {                                                                                                  \
        if (sp == 0) { finish(); break; }                                                                       \
        sp--;                                                                                                   \
        SEntry e_;                                                                                              \
        if (sp < ZR_LDS_LEVELS) e_ = lstack[sp * 64 + lbase]; else e_ = gstack[(size_t)(sp - ZR_LDS_LEVELS) * 64]; \
        if (e_.tn <= tbest_f) { cur = e_.node; pend_i = 0; st = (e_.node & X_LEAF_BIT) ? X_LEAF : X_NODE; break; } \
    }

// one child from the parametric distances of its six planes (absolute slack already inside): entry distance or +inf
#define ZR_SLAB(X0, X1, Y0, Y1, Z0, Z1, RF_IN, TN, RF)                                                      \
    {                                                                                                       \
        const float x0 = (X0), x1 = (X1), y0 = (Y0), y1 = (Y1), z0 = (Z0), z1 = (Z1);                       \
        float n_ = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.000999f));            \
        float f_ = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tbest_f));              \
        n_ = fmaf(fabsf(n_), -9.5367432e-7f, n_);                                                           \
        f_ = fmaf(fabsf(f_), 9.5367432e-7f, f_);                                                            \
        const bool empty_ = (RF_IN) == ZR_REF_EMPTY;                                                        \
        const bool hit_ = (n_ <= f_) && !empty_;                                                            \
        if (COUNT && !empty_) c_nodes++;                                                                    \
        TN = hit_ ? n_ : INFf;                                                                              \
        RF = (RF_IN);                                                                                       \
    }
// the slab constants of `ray` (see the header comment of this kernel)
#define ZR_RAY_CONSTANTS()                                                                                  \
    {                                                                                                       \
        const float NANf = __builtin_nanf("");                                                              \
        idx_ = 1.0f / (float)ray.d.x; idy_ = 1.0f / (float)ray.d.y; idz_ = 1.0f / (float)ray.d.z;            \
        float ocx = (float)(-ray.o.x * (double)idx_), ocy = (float)(-ray.o.y * (double)idy_), ocz = (float)(-ray.o.z * (double)idz_); \
        /* 2^100 / 2^120: far inside the float range, so that no product with a plane or a scale overflows */ \
        if (!(fabsf(idx_) < 1.2676506e30f) || !(fabsf(ocx) < 1.3292280e36f)) { idx_ = 0.0f; ocx = NANf; }    \
        if (!(fabsf(idy_) < 1.2676506e30f) || !(fabsf(ocy) < 1.3292280e36f)) { idy_ = 0.0f; ocy = NANf; }    \
        if (!(fabsf(idz_) < 1.2676506e30f) || !(fabsf(ocz) < 1.3292280e36f)) { idz_ = 0.0f; ocz = NANf; }    \
        /* the ENTRY plane of an axis (the lower one when id > 0) gets the smaller constant, the exit plane the larger one */ \
        const float sx = fabsf(ocx) * 2.3841858e-7f, sy = fabsf(ocy) * 2.3841858e-7f, sz = fabsf(ocz) * 2.3841858e-7f; \
        cnx = ocx - sx; cfx = ocx + sx; cny = ocy - sy; cfy = ocy + sy; cnz = ocz - sz; cfz = ocz + sz;      \
    }
// the root's FP32 planes (once per ray): lower / upper plane with the constant of the role it plays for this ray
#define ZR_FBOX(N, C, TN, RF)                                                                               \
    ZR_SLAB(fmaf((N).lox[C], idx_, idx_ > 0.0f ? cnx : cfx), fmaf((N).hix[C], idx_, idx_ > 0.0f ? cfx : cnx),  \
            fmaf((N).loy[C], idy_, idy_ > 0.0f ? cny : cfy), fmaf((N).hiy[C], idy_, idy_ > 0.0f ? cfy : cny),  \
            fmaf((N).loz[C], idz_, idz_ > 0.0f ? cnz : cfz), fmaf((N).hiz[C], idz_, idz_ > 0.0f ? cfz : cnz), (N).ref[C], TN, RF)
// the four children by entry distance (5-comparator network): push far -> near, continue with the nearest
#define ZR_DESCEND()                                                                                        \
    {                                                                                                       \
        cswap(tn0, r0, tn1, r1); cswap(tn2, r2, tn3, r3); cswap(tn0, r0, tn2, r2); cswap(tn1, r1, tn3, r3); cswap(tn1, r1, tn2, r2); \
        if (tn3 < INFf) ZR_PUSH(r3, tn3)                                                                    \
        if (tn2 < INFf) ZR_PUSH(r2, tn2)                                                                    \
        if (tn1 < INFf) ZR_PUSH(r1, tn1)                                                                    \
        if (tn0 < INFf) { cur = r0; pend_i = 0; st = (r0 & X_LEAF_BIT) ? X_LEAF : X_NODE; }                 \
        else { ZR_POP_NEXT() }                                                                              \
    }

    const unsigned long long iter_cap = (unsigned long long)B.P * 64ull + (1ull << 24);
    unsigned long long iter = 0;
#ifdef ZR_WAVE_PROFILE
    const unsigned long long t_begin = wall_clock64();
    unsigned long long p_exec[3] = {0, 0, 0}, p_lanes[3] = {0, 0, 0};
    // per-phase lane histogram: iterations of NODE / LEAF / FETCH by the number of lanes the phase ran with, in eight buckets of eight lanes
    __shared__ unsigned int p_hist[ST_EXT_GROUP][24];
    if (lane < 24) p_hist[threadIdx.x >> 6][lane] = 0;
#endif
    for (; iter < iter_cap; iter++) {
        const uint32_t lkind = (cur >> 28) & 7u;
        const int n1 = __popcll(__ballot(st == X_NODE));
        const int n2t = __popcll(__ballot(st == X_LEAF && lkind == ZR_PRIM_TRIANGLE));
        const int n2s = __popcll(__ballot(st == X_LEAF && lkind == ZR_PRIM_SPHERE));
        const int n2g = LEVEL > 0 ? __popcll(__ballot(st == X_LEAF)) - n2t - n2s : 0;
        const int n0 = work_left ? __popcll(__ballot(st == X_IDLE)) : 0;
#ifdef ST_LEAF_ALL
        const int n2 = n2t + n2s + n2g;
#else
        const int n2 = n2t > n2s ? (n2t > n2g ? n2t : n2g) : (n2s > n2g ? n2s : n2g);
#endif
        if (n1 + n2 + n0 == 0) break;

        if (n0 >= ST_FETCH_MIN || (n0 > 0 && n0 >= n1 && n0 >= n2)) {
            // ================= FETCH: idle lanes take the next ray indices =================
            // rays are handed out from a wave-private chunk; one global atomic per ST_CHUNK rays
            const unsigned long long idle = __ballot(st == X_IDLE);
            uint32_t n = (uint32_t)__popcll(idle);
#ifdef ZR_WAVE_PROFILE
            p_exec[2]++; p_lanes[2] += n;
            if (lane == 0 && n > 0) p_hist[threadIdx.x >> 6][16 + ((n - 1) >> 3)]++;
#endif
            while (chunk_next >= chunk_end && work_left) {
                // reserve the next chunk of this wave's shard; an exhausted shard sends the wave to the next one
                uint32_t nb = 0;
                if (lane == 0) nb = atomicAdd(&B.ctl[16 + 32 * head_shard + 8], 1u);
                nb = __builtin_amdgcn_readfirstlane(nb);
                const unsigned long long first = ((unsigned long long)nb * ST_SHARDS + head_shard) * ST_CHUNK;
                if (first < (unsigned long long)B.P) {
                    chunk_next = (uint32_t)first;
                    chunk_end = first + ST_CHUNK < (unsigned long long)B.P ? (uint32_t)first + ST_CHUNK : B.P;
                } else {
                    head_shard = (head_shard + 1) % ST_SHARDS;
                    if (++shards_tried >= ST_SHARDS) work_left = false;
                }
            }
            if (!work_left) n = 0;
            if (n > chunk_end - chunk_next) n = chunk_end - chunk_next;
            const uint32_t base = chunk_next;
            chunk_next += n;
            const uint32_t lim = base + n;
            if (st == X_IDLE) {
                const uint32_t my = base + (uint32_t)__popcll(idle & lt_mask);
                if (my < lim) {
                    // LEVEL > 0: meta word and ray requested together, one memory round trip per refill instead of two (an inactive
                    // slot's ray row is addressable like any other): cfg5 -1 %.  The lean build has no registers to hold the ray
                    // while the meta word is tested (2 more spilled, cfg3 +3 %): it asks for the ray once the slot is known active.
                    const uint2 m = B.ld2(SF_MA, my);
                    V3 ro_ = mk(0, 0, 0), rd_ = mk(0, 0, 0);
                    double key_ = 0;
                    if (LEVEL > 0) {
                        ro_ = B.ld3(SF_RAY, my); rd_ = B.ld3(SF_RAY + 3, my); key_ = B.ld(SF_KEY, my);
                        asm volatile("" ::"v"(ro_.x), "v"(rd_.z), "v"(key_));   // keeps the requests above the branch
                    }
                    if (m.y & F_ACTIVE) {
                        slot = my;
                        if (LEVEL > 0) { ray.o = ro_; ray.d = rd_; } else { ray.o = B.ld3(SF_RAY, my); ray.d = B.ld3(SF_RAY + 3, my); }
                        if (LEVEL > 0) { g.key = (uint64_t)__double_as_longlong(key_); g.bounce = m.y & 0xFFu; }
                        ZR_RAY_CONSTANTS()
                        tbest = INF; tbest_f = INFf; kbest = NONE; cur = 0; sp = 0; pend_i = 0; inst_cur = 0;
                        if (COUNT) c_seg++;
                        // the root's FP32 boxes come with the kernel arguments: no memory access for the first step, and a
                        // ray that misses the whole world is finished right here
                        float tn0, tn1, tn2, tn3;
                        uint32_t r0, r1, r2, r3;
                        ZR_FBOX(sc.root, 0, tn0, r0)
                        ZR_FBOX(sc.root, 1, tn1, r1)
                        ZR_FBOX(sc.root, 2, tn2, r2)
                        ZR_FBOX(sc.root, 3, tn3, r3)
                        ZR_DESCEND()
                    }
                }
            }
        } else if (n1 * ST_BIAS_NODE >= n2 * ST_BIAS_LEAF) {
            // ================= NODE: one 4-wide node per lane =================
            if (COUNT) { s_exec[0]++; s_lanes[0] += n1; }
#ifdef ZR_WAVE_PROFILE
            p_exec[0]++; p_lanes[0] += n1;
            if (lane == 0 && n1 > 0) p_hist[threadIdx.x >> 6][(n1 - 1) >> 3]++;
#endif
            if (st == X_NODE) {
                float tn0, tn1, tn2, tn3;
                uint32_t r0, r1, r2, r3;
#ifdef ST_DUMMY_VALU   /* measurement aid: ST_DUMMY_VALU extra vector instructions per node step (four independent chains) */
                {
                    float d0_ = idx_, d1_ = idy_, d2_ = idz_, d3_ = cnx;
#pragma unroll
                    for (int k_ = 0; k_ < ST_DUMMY_VALU / 4; k_++) {
                        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d0_)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d1_));
                        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d2_)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d3_));
                    }
                    asm volatile("" ::"v"(d0_), "v"(d1_), "v"(d2_), "v"(d3_));
                }
#endif
#ifdef ST_NODE_MINMAX   /* A/B: the min / max form of round 2 */
                const uint4* nq = reinterpret_cast<const uint4*>(sc.quads + cur);
                const uint4 w0 = nq[0], w1 = nq[1], w2 = nq[2], ref = nq[3];
                const bool px_ = idx_ > 0.0f, py_ = idy_ > 0.0f, pz_ = idz_ > 0.0f;
                // t = fmaf(q, a, b): a = scale * id, b = the node origin's parametric distance (the lower planes' b moved
                // by -2^-15 a, the upper planes' by +2^-15 a: towards "earlier" resp. "later" whatever the sign of id)
                const float ax_ = __uint_as_float(w0.w) * idx_, ay_ = __uint_as_float(w1.x) * idy_, az_ = __uint_as_float(w1.y) * idz_;
                const float blx = fmaf(ax_, -3.0517578e-5f, fmaf(__uint_as_float(w0.x), idx_, (px_ ? cnx : cfx))), bhx = fmaf(ax_, 3.0517578e-5f, fmaf(__uint_as_float(w0.x), idx_, (px_ ? cfx : cnx)));
                const float bly = fmaf(ay_, -3.0517578e-5f, fmaf(__uint_as_float(w0.y), idy_, (py_ ? cny : cfy))), bhy = fmaf(ay_, 3.0517578e-5f, fmaf(__uint_as_float(w0.y), idy_, (py_ ? cfy : cny)));
                const float blz = fmaf(az_, -3.0517578e-5f, fmaf(__uint_as_float(w0.z), idz_, (pz_ ? cnz : cfz))), bhz = fmaf(az_, 3.0517578e-5f, fmaf(__uint_as_float(w0.z), idz_, (pz_ ? cfz : cnz)));
#define ZR_QBOX(C, RF_IN, TN, RF)                                                                                          \
    ZR_SLAB(fmaf((float)((w1.z >> (8 * C)) & 0xFFu), ax_, blx), fmaf((float)((w2.y >> (8 * C)) & 0xFFu), ax_, bhx),         \
            fmaf((float)((w1.w >> (8 * C)) & 0xFFu), ay_, bly), fmaf((float)((w2.z >> (8 * C)) & 0xFFu), ay_, bhy),         \
            fmaf((float)((w2.x >> (8 * C)) & 0xFFu), az_, blz), fmaf((float)((w2.w >> (8 * C)) & 0xFFu), az_, bhz), RF_IN, TN, RF)
                ZR_QBOX(0, ref.x, tn0, r0)
                ZR_QBOX(1, ref.y, tn1, r1)
                ZR_QBOX(2, ref.z, tn2, r2)
                ZR_QBOX(3, ref.w, tn3, r3)
#undef ZR_QBOX
#else
                const uint4* nq = reinterpret_cast<const uint4*>(sc.quads + cur);
                const uint4 w0 = nq[0], w1 = nq[1], w2 = nq[2], ref = nq[3];
                // t = fmaf(q, a, b): a = scale * id, b = the node origin's parametric distance (the entry planes' b moved by -2^-15 |a|,
                // the exit planes' by +2^-15 |a|).  The sign of id says which of an axis' two planes the ray enters through — for all four
                // children at once: the word of lower planes or the word of upper planes — so no per-child min / max is needed, and the
                // children are evaluated two at a time (v_pk_fma_f32): the planes' values are those of the min / max form, bit for bit.
                typedef float f2_ __attribute__((ext_vector_type(2)));
                const float ax_ = __uint_as_float(w0.w) * idx_, ay_ = __uint_as_float(w1.x) * idy_, az_ = __uint_as_float(w1.y) * idz_;
                const float bnx = fmaf(fabsf(ax_), -3.0517578e-5f, fmaf(__uint_as_float(w0.x), idx_, cnx)), bfx = fmaf(fabsf(ax_), 3.0517578e-5f, fmaf(__uint_as_float(w0.x), idx_, cfx));
                const float bny = fmaf(fabsf(ay_), -3.0517578e-5f, fmaf(__uint_as_float(w0.y), idy_, cny)), bfy = fmaf(fabsf(ay_), 3.0517578e-5f, fmaf(__uint_as_float(w0.y), idy_, cfy));
                const float bnz = fmaf(fabsf(az_), -3.0517578e-5f, fmaf(__uint_as_float(w0.z), idz_, cnz)), bfz = fmaf(fabsf(az_), 3.0517578e-5f, fmaf(__uint_as_float(w0.z), idz_, cfz));
                const bool px_ = idx_ > 0.0f, py_ = idy_ > 0.0f, pz_ = idz_ > 0.0f;
                const uint32_t qnx = px_ ? w1.z : w2.y, qfx = px_ ? w2.y : w1.z;
                const uint32_t qny = py_ ? w1.w : w2.z, qfy = py_ ? w2.z : w1.w;
                const uint32_t qnz = pz_ ? w2.x : w2.w, qfz = pz_ ? w2.w : w2.x;
#define ZR_Q2(W, S) ((f2_){(float)(((W) >> (S)) & 0xFFu), (float)(((W) >> ((S) + 8)) & 0xFFu)})
#define ZR_T2(W, S, A, Bc) __builtin_elementwise_fma(ZR_Q2(W, S), ((f2_){(A), (A)}), ((f2_){(Bc), (Bc)}))
                const f2_ nx01 = ZR_T2(qnx, 0, ax_, bnx), nx23 = ZR_T2(qnx, 16, ax_, bnx), fx01 = ZR_T2(qfx, 0, ax_, bfx), fx23 = ZR_T2(qfx, 16, ax_, bfx);
                const f2_ ny01 = ZR_T2(qny, 0, ay_, bny), ny23 = ZR_T2(qny, 16, ay_, bny), fy01 = ZR_T2(qfy, 0, ay_, bfy), fy23 = ZR_T2(qfy, 16, ay_, bfy);
                const f2_ nz01 = ZR_T2(qnz, 0, az_, bnz), nz23 = ZR_T2(qnz, 16, az_, bnz), fz01 = ZR_T2(qfz, 0, az_, bfz), fz23 = ZR_T2(qfz, 16, az_, bfz);
#undef ZR_T2
#undef ZR_Q2
#define ZR_QBOX(NX, NY, NZ, FX, FY, FZ, RF_IN, TN, RF)                                                      \
    {                                                                                                       \
        float n_ = fmaxf(fmaxf(NX, NY), fmaxf(NZ, 0.000999f));                                              \
        float f_ = fminf(fminf(FX, FY), fminf(FZ, tbest_f));                                                \
        n_ = fmaf(fabsf(n_), -9.5367432e-7f, n_);                                                           \
        f_ = fmaf(fabsf(f_), 9.5367432e-7f, f_);                                                            \
        const bool empty_ = (RF_IN) == ZR_REF_EMPTY;                                                        \
        const bool hit_ = (n_ <= f_) && !empty_;                                                            \
        if (COUNT && !empty_) c_nodes++;                                                                    \
        TN = hit_ ? n_ : INFf;                                                                              \
        RF = (RF_IN);                                                                                       \
    }
                ZR_QBOX(nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x, ref.x, tn0, r0)
                ZR_QBOX(nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y, ref.y, tn1, r1)
                ZR_QBOX(nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x, ref.z, tn2, r2)
                ZR_QBOX(nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y, ref.w, tn3, r3)
#undef ZR_QBOX
#endif
                ZR_DESCEND()
            }
        } else {
            // ================= LEAF: one primitive per lane, the kind with most waiting lanes =================
            if (COUNT) { s_exec[1]++; s_lanes[1] += n2; }
#ifdef ZR_WAVE_PROFILE
            p_exec[1]++; p_lanes[1] += n2;
            if (lane == 0 && n2 > 0) p_hist[threadIdx.x >> 6][8 + ((n2 - 1) >> 3)]++;
#endif
            const bool is_leaf = st == X_LEAF;
#ifdef ST_LEAF_ALL
            // experiment: every leaf kind that has waiting lanes is served in this iteration (kind by kind, each under a wave-uniform
            // guard) instead of only the kind with most lanes
            const bool do_tri = n2t > 0;
            const bool do_sph = n2s > 0;
#else
            const bool do_tri = n2t == n2;
            const bool do_sph = !do_tri && n2s == n2;
#endif
            const uint32_t prim = (cur & 0xFFFFFFu) + pend_i;
            bool tested = false;
            if (do_tri) {
                if (is_leaf && lkind == ZR_PRIM_TRIANGLE) {
                    double t;
                    if (COUNT) c_tri++;
                    if (triangle_t(sc.tri_v + (size_t)prim * ZR_TRI_STRIDE, ray, 0.001, tbest, t)) {
                        tbest = t; tbest_f = __double2float_ru(t); ibest = prim;
                        kbest = (LEVEL == 3 && inst_cur) ? (ZR_KIND_INSTANCE | ((inst_cur - 1u) << 8)) : lkind;   // a triangle of a placed run: kind word = placement
                    }
                    tested = true;
                }
            }
#ifdef ST_LEAF_ALL
            if (do_sph) {
#else
            else if (do_sph) {
#endif
                if (is_leaf && lkind == ZR_PRIM_SPHERE) {
                    double t;
                    if (COUNT) c_sph++;
                    if (sphere_t(sc.spheres + (size_t)prim * 4, ray, 0.001, tbest, t)) { tbest = t; tbest_f = __double2float_ru(t); kbest = lkind; ibest = prim; }
                    tested = true;
                }
            }
#ifdef ST_LEAF_ALL
            if (LEVEL > 0 && is_leaf && lkind != ZR_PRIM_TRIANGLE && lkind != ZR_PRIM_SPHERE) {
#else
            else if (LEVEL > 0 && is_leaf && lkind != ZR_PRIM_TRIANGLE && lkind != ZR_PRIM_SPHERE) {
#endif
                double t;
                if (COUNT && lkind < ZR_KIND_INSTANCE) {
                    uint32_t kk = lkind;
                    if (kk == ZR_KIND_WRAPPED) kk = sc.wrapped[prim].type;
                    if (kk == ZR_PRIM_SPHERE) c_sph++; else if (kk == ZR_PRIM_TRIANGLE) c_tri++; else if (kk == ZR_PRIM_CUBE || kk == ZR_KIND_PCUBE) c_cube++; else c_med++;
                }
                bool h;
                if (LEVEL == 3 && lkind == ZR_KIND_INSTANCE) {
                    // ENTER a placed run of triangles (two-level BVH): a sentinel stays on the stack where the world's walk goes on, the
                    // ray is mapped through the placement's wrappers (t is the same in both spaces: the reference's wrappers do not
                    // renormalise the direction) and the lane continues in the run's own 4-wide tree — same loop, same stack
                    const DInstance in = sc.insts[prim];
                    ZR_PUSH(ZR_REF_LEAF | (7u << 28), -INFf)
                    ray = chain_ray(sc, in.chain_first, in.chain_count, ray);
                    ZR_RAY_CONSTANTS()
                    inst_cur = prim + 1u;
                    cur = in.pad_; pend_i = 0; st = X_NODE;
                } else if (LEVEL == 3 && lkind == 7u) {
                    // LEAVE: the run's tree is exhausted (its entries were above the sentinel); the world ray again, from the slot
                    ray.o = B.ld3(SF_RAY, slot); ray.d = B.ld3(SF_RAY + 3, slot);
                    ZR_RAY_CONSTANTS()
                    inst_cur = 0;
                    ZR_POP_NEXT()
                } else {
                if (LEVEL == 1) {   // cubes, placed cubes, plain media: no op-list interpreter in this build
                    if (lkind == ZR_KIND_PCUBE) h = pcube_t(sc.pcubes + (size_t)prim * ZR_PCUBE_STRIDE, ray, 0.001, tbest, t);
                    else if (lkind == ZR_PRIM_CUBE) h = cube_t(sc.cubes + (size_t)prim * 6, ray, 0.001, tbest, t);
                    else h = medium_plain_t(sc, prim, ray, 0.001, tbest, g, t);
                } else h = object_t(sc, lkind, prim, ray, 0.001, tbest, g, t);
                if (h) { tbest = t; tbest_f = __double2float_ru(t); kbest = lkind; ibest = prim; }
                tested = true;
                }
            }
            if (tested) {
                pend_i++;
                if (pend_i > ((cur >> 24) & 0xFu)) { ZR_POP_NEXT() }
            }
        }
    }
    if (iter >= iter_cap && lane == 0) atomicAdd(&B.ctl[2], 1u);
#ifdef ZR_WAVE_PROFILE
    if (!COUNT && lane == 0) {  // development build: wave lifetime (10 ns ticks) and phase statistics into the raw counter words
        atomicAdd(&gctr[13], wall_clock64() - t_begin); atomicAdd(&gctr[14], 1ull);
        for (int k = 0; k < 3; k++) { atomicAdd(&gctr[1 + 2 * k], p_exec[k]); atomicAdd(&gctr[2 + 2 * k], p_lanes[k]); }
        for (int k = 0; k < 24; k++) if (p_hist[threadIdx.x >> 6][k]) atomicAdd(&gctr[16 + k], (unsigned long long)p_hist[threadIdx.x >> 6][k]);   // (the context's counter block has 48 words)
    }
#endif
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
// MODE 0: the render.  MODE 1 / 2: the two passes of the reflection / refraction split (camera.hpp:490-517) on the same pipeline:
// 1 = the beauty pass, which also records per unit how many draws and segments its path consumed; 2 = the replay: the camera
// ray is traced again, at its first hit the sample's stream is positioned after the beauty path (so the SECOND scatter of that
// hit gets the draws the reference gives it), the scattered path runs as ray_color(scattered, max_depth - 1), and its
// luma-clamped radiance times the attenuation is written with its class.
#ifndef ST_SHADE_WAVES_LEAN
#define ST_SHADE_WAVES_LEAN 6
#endif
template <bool COUNT, int MODE, bool LEAN = false>
__global__ __launch_bounds__(256, LEAN ? ST_SHADE_WAVES_LEAN : ST_SHADE_WAVES) void stream_shade(DScene sc, DCamera cam, DEnv env, uint64_t seed, StreamBuf B,
                                                    unsigned long long* __restrict__ gctr) {
    const uint32_t slot0 = blockIdx.x * 256 + threadIdx.x;
    if (slot0 < ST_SHARDS) B.ctl[16 + 32 * slot0 + 8] = 0;  // EXTEND of the next round starts from chunk 0 of every shard
    // the split passes count through LDS into a per-block record (a global atomic per thread costs ~1.5 ms per round and word)
    __shared__ unsigned long long s_cnt[4];
    if (MODE != 0 && threadIdx.x < 4) s_cnt[threadIdx.x] = 0;   // ordered before its use by the partition's barriers
    // In-block MATERIAL SORT: the 256 slots of this block are re-dealt to its threads by what their segment will execute —
    // class 0..4 = the hit's material kind (lambertian, metal, dielectric, light, isotropic: material.hpp:58-279,
    // constant_medium.hpp:14-18), 5 = a hit without a valid material, 6 = miss (background), 7 = inactive slot — so that a
    // wave runs one scatter routine (and one rejection sampler) under a full exec mask instead of five under sparse ones;
    // unsorted, almost every wave of a mixed scene runs every branch.  A counting sort through LDS: per-wave ballots per class,
    // a 4 x 8 table of counts, ranks by prefix popcount.  The material lookup it needs (primitive -> material id -> kind) is
    // the first access of the lines the hit record reads anyway; a scene with a single material kind skips it (hit / miss only).
    // The whole kernel is a chain of dependent memory round trips (73 % of its wave cycles sit in s_waitcnt, SQ counters of
    // round 2), so the rows a segment needs are requested as early and as much at once as their addresses are known:
    // the slot's meta word and its hit together; the sort hands both through LDS to the thread that will shade the slot (no
    // second read); that thread requests every state row of the segment in one go.
    __shared__ unsigned int wave_cls[4][8];
    __shared__ uint4 x_mk[256];          // (meta.x, meta.y, hit kind, hit index) of the slot a thread will shade
    __shared__ unsigned char x_src[256]; // ... and which slot of the block that is
    uint32_t src;   // which slot of the block this thread shades
    uint2 m, ki;
    {
        uint2 m0; m0.x = 0; m0.y = 0;
        uint2 k0; k0.x = 0xFFFFFFFFu; k0.y = 0;
        if (slot0 < B.P) { const SlotAt S0 = slot_at(B, blockIdx.x, threadIdx.x); m0 = S0.ld2(SF_MA); k0 = S0.ld2(SF_HIT_KI); }
#ifdef ZR_SHADE_NO_PARTITION
        src = threadIdx.x; m = m0; ki = k0;
#else
        uint32_t cls = 7;
        if (m0.y & F_ACTIVE) {
            if (k0.x == 0xFFFFFFFFu) cls = 6;
            else {
                cls = 0;
#ifndef ZR_SHADE_HITMISS_ONLY
                if (sc.mat_kinds & (sc.mat_kinds - 1u)) {   // more than one material kind in the scene (wave-uniform)
                    const uint32_t mat = object_material(sc, k0.x, k0.y);
                    cls = mat < sc.n_mats ? sc.mats[mat].kind : 5u;
                    if (cls > 5u) cls = 5u;
                }
#endif
            }
        }
        const int w = threadIdx.x >> 6, wl = threadIdx.x & 63;
        const unsigned long long below = (1ull << wl) - 1ull;
        unsigned int my_rank = 0;
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) {
            const unsigned long long bm = __ballot(cls == c);
            if (wl == 0) wave_cls[w][c] = (unsigned int)__popcll(bm);
            if (cls == c) my_rank = (unsigned int)__popcll(bm & below);
        }
        __syncthreads();
        unsigned int base = 0;   // slots of lower classes in any wave + slots of this class in earlier waves
        for (uint32_t c = 0; c < 8; c++)
            for (int k = 0; k < 4; k++)
                if (c < cls || (c == cls && k < w)) base += wave_cls[k][c];
        uint4 mk4; mk4.x = m0.x; mk4.y = m0.y; mk4.z = k0.x; mk4.w = k0.y;
        x_mk[base + my_rank] = mk4;
        x_src[base + my_rank] = (unsigned char)threadIdx.x;
        __syncthreads();
        mk4 = x_mk[threadIdx.x];
        src = x_src[threadIdx.x];
        m.x = mk4.x; m.y = mk4.y; ki.x = mk4.z; ki.y = mk4.w;
#endif
    }
    const SlotAt S = slot_at(B, blockIdx.x, src);
    bool active_after = false, want_unit = false;
    uint32_t c_samp = 0, c_seg2 = 0, c_hit2 = 0; unsigned long long c_draws = 0;
    {
        if (m.y & F_ACTIVE) {   // (a thread beyond the pool got meta = 0 from the prologue)
            const uint32_t NONE = 0xFFFFFFFFu;
            const bool first = (m.y & F_FIRST) != 0;
#ifdef ZR_SHADE_TOUCH
            // experiment, off: first and last line of the record object_rec() will read, requested with the state rows (cfg3 +1.3 %:
            // the record's own read costs less than the extra requests).  The loads
            // are invisible to the compiler's wait counting, which is safe: memory returns in order, so every wait it inserts for
            // a younger load covers them; the register they write stays reserved until the hit branch has waited for the ray.
            uint32_t touch = 0;
            if (ki.x == ZR_PRIM_TRIANGLE) {
                const double* q = sc.tri_s + (size_t)ki.y * 20;
                asm volatile("global_load_dword %0, %1, off\n\tglobal_load_dword %0, %1, off offset:156" : "=&v"(touch) : "v"(q) : "memory");
            } else if (ki.x == ZR_PRIM_SPHERE) {
                const double* q = sc.spheres + (size_t)ki.y * 4;
                asm volatile("global_load_dword %0, %1, off" : "=&v"(touch) : "v"(q) : "memory");
            }
#endif
            // every row this segment reads, requested in one go (a row of a slot is always addressable; what a path does not
            // need is not requested: beta before the second hit, the unit id is 8 bytes)
            Ray ray; ray.o = S.ld3(SF_RAY); ray.d = S.ld3(SF_RAY + 3);
            Rng g; g.key = (uint64_t)__double_as_longlong(S.ld(SF_KEY)); g.k = m.x; g.bounce = (m.y & 0xFFu) + 1u;  // this query is complete
            const uint2 mb_now = S.ld2(SF_MB);
            const double t_hit = S.ld(SF_HIT_T);
            // after the first hit L = 0 and beta = 1 (camera.hpp:930): both stay implicit (two flag bits) until something else is
            // stored, which saves their 48 bytes written and read back per path; the values used are the same (0 + x, 1 * x)
            V3 beta_now = mk(1, 1, 1);
            if (!first && !(m.y & F_BONE)) beta_now = S.ld3(SF_BETA);
            V3 att0_now = mk(0, 0, 0);
            if (!first && ki.x == NONE) att0_now = S.ld3(SF_ATT0);   // a miss ends the path: its sample is att0 * (L + beta * background)
            int b_inner = (int)((m.y >> 8) & 0xFFu);
            const int depth_inner = cam.max_depth - 1;
            auto load_L = [&]() { return (m.y & F_LZERO) ? mk(0, 0, 0) : S.ld3(SF_L); };
            auto load_beta = [&]() { return beta_now; };
            uint32_t keep_lzero = m.y & F_LZERO;
            // the split passes count segments and hits here (SHADE sees every segment exactly once), so that their EXTEND can be
            // the uninstrumented build; the replay's first segment is the beauty pass's, found again: not counted
            if (COUNT && MODE != 0 && !(MODE == 2 && first)) { c_seg2++; if (ki.x != 0xFFFFFFFFu) c_hit2++; }
            // L, att0 and the slot's running sum are read only on the paths that need them
            bool ended = false;
            V3 contrib = mk(0, 0, 0);   // added to the slot sum when the path ends
            V3 add_now = mk(0, 0, 0);   // emission of a primary hit: added to the slot sum immediately
            bool has_add = false;
            bool no_output = false;     // MODE 2: the sample contributes nothing to the split frames
            if (ki.x == NONE) {
                if (MODE == 2 && first) { ended = true; no_output = true; }   // the primary ray saw the background only (camera.hpp:518-526)
                else {
                    V3 bg = background(sc, env, ray.d);
                    if (MODE == 2) {
                        V3 scol = load_L() + load_beta() * bg;
                        const double luma = 0.2126 * len(scol);                  // camera.hpp:499-503
                        if (luma > 2.0) scol = scol * (2.0 / luma);
                        contrib = att0_now * scol;
                    } else {
                        contrib = first ? bg : att0_now * (load_L() + load_beta() * bg);  // camera.hpp:520 / 941,1000
                    }
                    ended = true;
                }
            } else {
                const double t = t_hit;
                V3 em, att; Ray nr; bool sc_ok;
                uint32_t cls_now = 0;
                if (LEAN) {
                    RecL rec;
                    lean_rec(sc, ki.x, ki.y, ray, t, rec);
                    sc_ok = lean_shade(sc, ray, rec, em, att, nr, g);
                } else {
                    Rec rec;
#ifdef ZR_SHADE_TOUCH
                    asm volatile("" ::"v"(touch), "v"(ray.o.x), "v"(ray.d.z));   // the ray is here, so the touches (older) have returned
#endif
                    object_rec(sc, ki.x, ki.y, ray, t, rec);
                    em = emitted(sc, rec);
                    if (MODE == 2 && first) {   // the stream continues where the beauty path of this sample stopped
                        const uint2 ke = B.kend[mb_now.x];
                        g.k = ke.x; g.bounce = ke.y;
                    }
                    sc_ok = scatter(sc, ray, rec, att, nr, g);
                    if (MODE == 2 && first && sc_ok) {   // camera.hpp:506-516
                        const V3 reflected_dir = reflect(unit(ray.d), unit(rec.n));
                        if (dot(unit(nr.d), reflected_dir) > 0.9) cls_now = 1;
                        else if (dot(nr.d, rec.n) < 0) cls_now = 2;
                    }
                }
                const bool has_em = em.x != 0.0 || em.y != 0.0 || em.z != 0.0;
                if (first) {  // ray_color_from_hit, camera.hpp:989-1004
                    if (MODE != 2 && has_em) { add_now = em; has_add = true; }
                    if (!sc_ok || depth_inner <= 0) { ended = true; if (MODE == 2) no_output = true; }
                    else {
                        S.st3(SF_ATT0, att);   // L = 0, beta = 1: implicit (F_LZERO | F_BONE below)
                        if (MODE == 2) { uint2 mb = mb_now; mb.y = cls_now; S.st2(SF_MB, mb); }
                        b_inner = 0;
                    }
                } else {      // body of ray_color's loop, camera.hpp:944-983
                    V3 beta = load_beta();
                    V3 L = mk(0, 0, 0);
                    bool have_L = false;
                    if (has_em) { L = load_L() + beta * em; have_L = true; }
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
                    if (stop) {
                        if (!have_L) L = load_L();
                        if (MODE == 2) { const double luma = 0.2126 * len(L); if (luma > 2.0) L = L * (2.0 / luma); }
                        contrib = S.ld3(SF_ATT0) * L; ended = true;
                    } else {
                        if (have_L) { S.st3(SF_L, L); keep_lzero = 0; }
                        S.st3(SF_BETA, beta);
                    }
                }
                if (!ended) {
                    // the path continues: publish the scattered ray (a primary hit's emission waits in SF_SUM)
                    S.st3(SF_RAY, nr.o); S.st3(SF_RAY + 3, nr.d);
                    if (has_add) S.st3(SF_SUM, add_now);
                    m.x = (uint32_t)g.k;
                    m.y = (g.bounce & 0xFFu) | ((uint32_t)b_inner << 8) | F_ACTIVE | (first ? (has_add ? F_L0 : 0u) : (m.y & F_L0)) |
                          (first ? (F_LZERO | F_BONE) : keep_lzero);
                    S.st2(SF_MA, m);
                    active_after = true;
                }
            }
            if (ended) {
                // radiance of this sample = L0 + att0 * L (camera.hpp:1000), written exactly once
                V3 rad = contrib;
                if (has_add) rad = add_now + rad;
                else if (MODE != 2 && !first && (m.y & F_L0)) rad = S.ld3(SF_SUM) + rad;
                const uint2 mb = mb_now;
                const uint32_t unit = mb.x;
                if (MODE != 2 || !no_output) {
                    double* pp = B.samples + (size_t)unit * 3;
                    pp[0] = rad.x; pp[1] = rad.y; pp[2] = rad.z;
                }
                if (MODE == 1) { uint2 ke; ke.x = (uint32_t)g.k; ke.y = g.bounce; B.kend[unit] = ke; }
                if (MODE == 2 && !no_output) B.cls[unit] = (unsigned char)mb.y;
                if (COUNT) {
                    if (MODE != 2) c_draws += g.k;
                    else if (!(first && ki.x == NONE)) c_draws += g.k - B.kend[unit].x;   // draws of the second path only
                }
                want_unit = true;
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
            if (wl == (int)__builtin_ctzll(wm)) k0 = atomicAdd(&B.uctl[32 * shard], (unsigned int)__popcll(wm));
            k0 = __shfl(k0, (int)__builtin_ctzll(wm), 64);
            if (want_unit) {
                const unsigned long long k = (unsigned long long)k0 + (unsigned long long)__popcll(wm & ((1ull << wl) - 1ull));
                const unsigned long long u = st_unit_of(B.unit_chunk, B.unit_base, shard, (unsigned long long)B.shard_k0 + k);
                if (u < (unsigned long long)B.n_units) { begin_sample(B, cam, seed, S, (uint32_t)u, c_samp); active_after = true; }
                else { uint2 z; z.x = 0; z.y = 0; S.st2(SF_MA, z); }
            }
        }
    }
    // active-slot count for the host's round loop: sharded like the other counters — one contended word would cost
    // ~3 ms per 16 M-slot round (262 144 wave atomics at ~90 per microsecond)
    const unsigned long long am = __ballot(active_after);
    if ((threadIdx.x & 63) == 0 && am != 0ull) atomicAdd(&B.ctl[16 + 32 * (blockIdx.x % ST_SHARDS) + 16], (unsigned int)__popcll(am));
    if (COUNT && MODE == 0) {
        if (c_samp) atomicAdd(&gctr[0], (unsigned long long)c_samp);
        if (c_draws) atomicAdd(&gctr[8], c_draws);
    }
    if (MODE != 0) {
        if (c_samp) atomicAdd(&s_cnt[0], (unsigned long long)c_samp);
        if (c_seg2) atomicAdd(&s_cnt[1], (unsigned long long)c_seg2);
        if (c_hit2) atomicAdd(&s_cnt[2], (unsigned long long)c_hit2);
        if (c_draws) atomicAdd(&s_cnt[3], c_draws);
        __syncthreads();
        if (threadIdx.x < 4 && s_cnt[threadIdx.x]) B.cpart[(size_t)blockIdx.x * 4 + threadIdx.x] += s_cnt[threadIdx.x];
    }
}

// ---- FUSED small-scene path ---------------------------------------------------------------------------------------------------
// A world of a few dozen objects (cfg5: seven placed cubes, a sphere, a medium) needs no tree and no pool: streaming its paths
// through EXTEND and SHADE moved 405 bytes of path state per segment through HBM for nine objects (profiles/r2_cfg5_traffic.json:
// 979 GB per frame), and the frame's 486 ms were those round trips.  Here a lane keeps its whole path in registers and tests
// EVERY leaf object of the world each segment — the loop counter is wave-uniform, so the objects' records arrive through scalar
// loads, once per wave — then shades the winner in place; a lane whose path ends writes the sample and starts the next work unit
// (persistent threads with regeneration: the intersection loop always runs under a full exec mask, only the material branch
// diverges).  HBM sees 24 bytes per SAMPLE (samples[unit], reduced by stream_reduce as in the pipeline): the kernel is bound by
// FP64 vector issue.  Arithmetic and decisions are the pipeline's (the same zr_device.h routines in the same order), so a frame
// agrees with the pipeline's to the last bits of FMA contraction.
// Work units: waves reserve chunks of ST_FUSED_CHUNK consecutive units (one atomic per chunk, ST_SHARDS interleaved counters), so
// that the 64 lanes of a wave share a pixel; every wave reaches the exit: it leaves when its lanes are idle and every shard is dry.
#ifndef ST_FUSED_CHUNK
#define ST_FUSED_CHUNK 256
#endif
#ifndef ST_FUSED_WAVES
#define ST_FUSED_WAVES 2
#endif
struct FusedBuf {
    const uint32_t* pixels; double* samples; unsigned int* uctl;   // uctl[32 s]: chunk counter of shard s
    uint32_t spp, n_units;
    uint32_t chunk_lo, chunk_hi;   // this launch hands out the chunks [chunk_lo, chunk_hi) of ST_FUSED_CHUNK units (a frame in one launch, or in
                                   // parts when the caller polls for cancellation / progress between them)
};
// plain medium from a kernel-argument record (medium_plain_t's arithmetic: constant_medium.hpp:39-77 on a bare boundary)
__device__ __forceinline__ bool medium_rec_t(const double* q, const Ray& r, double tmin, double tmax, const Rng& g, double& t) {
    double t1, t2;
    const double inf = __builtin_huge_val();
    const uint32_t btype = (uint32_t)__double_as_longlong(q[8]);
    if (btype == ZR_PRIM_SPHERE) {
        if (!sphere_t(q, r, -inf, inf, t1)) return false;
        if (!sphere_t(q, r, t1 + 0.0001, inf, t2)) return false;
    } else {
        if (!cube_t(q, r, -inf, inf, t1)) return false;
        if (!cube_t(q, r, t1 + 0.0001, inf, t2)) return false;
    }
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < 0) t1 = 0;
    const double rl = len(r.d);
    const double inside = (t2 - t1) * rl;
    const double xi = zr_bits_to_unit(zr_medium_bits(g.key, g.bounce, (uint32_t)__double_as_longlong(q[7])));
    const double hd = q[6] * log(xi);
    if (hd > inside) return false;
    t = t1 + hd / rl;
    return true;
}
template <int LEVEL, bool COUNT>
__device__ __forceinline__ void brute_hit(const DScene& sc, const FusedObjs& fo, const Ray& ray, const Rng& g, double& tbest, uint32_t& kbest, uint32_t& ibest, uint32_t* cn) {
    const double INF = __builtin_huge_val();
    tbest = INF; kbest = 0xFFFFFFFFu; ibest = 0;
    double t;
    const V3 inv_d = mk(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);   // for the cubes that see the world ray's direction (cube_t_inv)
    // the table's objects: wave-uniform loop counter and kind, records through scalar loads
    for (uint32_t i = 0; i < fo.n; i++) {
        const uint32_t k = fo.kind[i];
        const double* q = fo.rec[i];
        bool h;
        if (k == ZR_PRIM_SPHERE) h = sphere_t(q, ray, 0.001, tbest, t);
        else if (k == ZR_PRIM_TRIANGLE) h = triangle_t(q, ray, 0.001, tbest, t);
        else if (k == ZR_PRIM_CUBE) h = cube_t_inv(q, ray, inv_d, 0.001, tbest, t);
        else if (k == ZR_KIND_PCUBE) h = pcube_t_inv<false>(q, ray, inv_d, 0.001, tbest, t);   // (no scaled placed cube reaches this kernel: zr_commit.cpp finish_commit)
        else h = medium_rec_t(q, ray, 0.001, tbest, g, t);
        if (h) { tbest = t; kbest = k; ibest = fo.index[i]; }
        if (COUNT) { if (k == ZR_PRIM_SPHERE) cn[0]++; else if (k == ZR_PRIM_TRIANGLE) cn[1]++; else if (k == ZR_PRIM_CUBE || k == ZR_KIND_PCUBE) cn[2]++; else cn[3]++; }
    }
    if (LEVEL >= 2) {   // what the table cannot hold: media with a wrapped boundary, objects under wrapper chains (the op-list interpreter)
        for (uint32_t i = 0; i < sc.leaf_cnt[ZR_PRIM_MEDIUM]; i++) {
            if (sc.media[i].chain_count == 0) continue;   // (plain ones are in the table)
            if (medium_t(sc, i, ray, 0.001, tbest, g, t)) { tbest = t; kbest = ZR_PRIM_MEDIUM; ibest = i; }
            if (COUNT) cn[3]++;
        }
        for (uint32_t i = 0; i < sc.leaf_cnt[ZR_KIND_WRAPPED]; i++) {
            if (object_t(sc, ZR_KIND_WRAPPED, i, ray, 0.001, tbest, g, t)) { tbest = t; kbest = ZR_KIND_WRAPPED; ibest = i; }
            if (COUNT) { const uint32_t kk = sc.wrapped[i].type; if (kk == ZR_PRIM_SPHERE) cn[0]++; else if (kk == ZR_PRIM_TRIANGLE) cn[1]++; else if (kk == ZR_PRIM_CUBE) cn[2]++; else cn[3]++; }
        }
    }
}

template <int LEVEL, bool COUNT>
__global__ __launch_bounds__(256, ST_FUSED_WAVES) void fused_render(DScene sc, DCamera cam, DEnv env, uint64_t seed, FusedBuf B, unsigned long long* __restrict__ gctr, FusedObjs fo) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t NONE = 0xFFFFFFFFu;
    const int depth_inner = cam.max_depth - 1;
    // the lane's path
    bool active = false, first = true;
    uint32_t unit = 0;
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
    Rng g; g.key = 0; g.k = 0; g.bounce = 0;
    V3 L0 = mk(0, 0, 0), att0 = mk(0, 0, 0), L = mk(0, 0, 0), beta = mk(1, 1, 1);
    int b_inner = 0;
    // the wave's chunk of work units
    uint32_t chunk_next = 0, chunk_end = 0, shard = wave_id % ST_SHARDS, shards_tried = 0;
    bool units_left = true;
    uint32_t cn[4] = {0, 0, 0, 0}, c_samp = 0, c_seg = 0, c_hit = 0;
    unsigned long long c_draws = 0;
    const unsigned long long n_chunks = B.chunk_hi;
    for (;;) {
        // ---- regeneration: idle lanes take the wave's next units
        const unsigned long long idle = __ballot(!active);
        if (idle != 0ull && units_left) {
            uint32_t want = (uint32_t)__popcll(idle);
            while (chunk_next >= chunk_end && units_left) {
                uint32_t nb = 0;
                if (lane == 0) nb = atomicAdd(&B.uctl[32 * shard], 1u);
                nb = __builtin_amdgcn_readfirstlane(nb);
                const unsigned long long c = (unsigned long long)B.chunk_lo + (unsigned long long)nb * ST_SHARDS + shard;   // chunk index
                if (c < n_chunks) {
                    chunk_next = (uint32_t)(c * ST_FUSED_CHUNK);
                    const unsigned long long e = c * ST_FUSED_CHUNK + ST_FUSED_CHUNK;
                    chunk_end = e < (unsigned long long)B.n_units ? (uint32_t)e : B.n_units;
                } else {
                    shard = (shard + 1) % ST_SHARDS;
                    if (++shards_tried >= ST_SHARDS) units_left = false;
                }
            }
            if (!units_left) want = 0;
            if (want > chunk_end - chunk_next) want = chunk_end - chunk_next;
            const uint32_t base = chunk_next;
            chunk_next += want;
            if (!active) {
                const uint32_t my = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (my < want) {
                    unit = base + my;
                    const uint32_t pix_i = unit / B.spp, sample = unit - pix_i * B.spp;
                    const uint32_t pk = B.pixels[pix_i];
                    const int px = (int)(pk & 0xFFFFu), py = (int)(pk >> 16);
                    g.key = zr_stream_key(seed, (uint64_t)py * (uint64_t)cam.W + (uint64_t)px, (uint64_t)sample); g.k = 0; g.bounce = 0;
                    ray = camera_ray(cam, px, py, g);
                    first = true; active = true;
                    if (COUNT) c_samp++;
                }
            }
        }
        if (__ballot(active) == 0ull) { if (!units_left) break; else continue; }
        // ---- one segment for every active lane: closest hit over all leaf objects, then the shading of the winner
        double t_hit; uint32_t kind, idx;
        brute_hit<LEVEL, COUNT>(sc, fo, ray, g, t_hit, kind, idx, cn);
        if (active) {
            g.bounce++;
            if (COUNT) { c_seg++; if (kind != NONE) c_hit++; }
            bool ended = false;
            V3 rad = mk(0, 0, 0);
            if (kind == NONE) {
                const V3 bg = background(sc, env, ray.d);
                rad = first ? bg : L0 + att0 * (L + beta * bg);   // camera.hpp:520 / 941, 1000
                ended = true;
            } else {
                Rec rec;
                object_rec<false>(sc, kind, idx, ray, t_hit, rec);
                const V3 em = emitted(sc, rec);
                V3 att; Ray nr;
                const bool sc_ok = scatter(sc, ray, rec, att, nr, g);
                if (first) {   // ray_color_from_hit, camera.hpp:989-1004
                    L0 = em;
                    if (!sc_ok || depth_inner <= 0) { rad = L0; ended = true; }
                    else { att0 = att; L = mk(0, 0, 0); beta = mk(1, 1, 1); b_inner = 0; first = false; ray = nr; }
                } else {       // body of ray_color's loop, camera.hpp:944-983
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
                    if (stop) { rad = L0 + att0 * L; ended = true; } else ray = nr;
                }
            }
            if (ended) {
                double* pp = B.samples + (size_t)unit * 3;
                pp[0] = rad.x; pp[1] = rad.y; pp[2] = rad.z;
                if (COUNT) c_draws += g.k;
                active = false;
            }
        }
    }
    if (COUNT) {
        atomicAdd(&gctr[0], (unsigned long long)c_samp); atomicAdd(&gctr[1], (unsigned long long)c_seg);
        atomicAdd(&gctr[3], (unsigned long long)cn[0]); atomicAdd(&gctr[4], (unsigned long long)cn[1]); atomicAdd(&gctr[5], (unsigned long long)cn[2]); atomicAdd(&gctr[6], (unsigned long long)cn[3]);
        atomicAdd(&gctr[7], (unsigned long long)c_hit); atomicAdd(&gctr[8], c_draws);
    }
}

// THE DRAIN.  Once the last work unit has been handed out a frame lives on its longest paths: cfg5 (depth 50) spends 36 of its
// 118 rounds on a pool that is almost empty, and a round costs 0.8 ms however few paths it carries (every EXTEND wave still
// walks its share of the slot pool to find them).  So when no unit is left and fewer than 1/16 of the slots are active, the
// host has the survivors MOVED to the front of a small pool of their own (all 23 state rows of a slot: a slot is self-contained,
// its unit id travels with it) and runs the remaining rounds on that.  Which slot holds a path changes; nothing else does.
__global__ __launch_bounds__(256) void stream_compact(StreamBuf S, StreamBuf D, unsigned int* __restrict__ counter) {
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    bool act = false;
    if (slot < S.P) act = (S.ld2(SF_MA, slot).y & F_ACTIVE) != 0;
    const unsigned long long bm = __ballot(act);
    if (bm == 0ull) return;
    const int wl = threadIdx.x & 63, lead = (int)__builtin_ctzll(bm);
    uint32_t base = 0;
    if (wl == lead) base = atomicAdd(counter, (unsigned int)__popcll(bm));
    base = __shfl(base, lead, 64);
    if (!act) return;
    const uint32_t d = base + (uint32_t)__popcll(bm & ((1ull << wl) - 1ull));
    if (d >= D.P) return;   // (cannot happen: the host sized D from the same count)
#pragma unroll
    for (int f = 0; f <= SF_MB; f++) D.st(f, d, S.ld(f, slot));   // the 23 rows in use (the 24th is spare)
}
__global__ __launch_bounds__(256) void stream_compact_pad(StreamBuf D, uint32_t from) {   // the slots behind the survivors: inactive
    const uint32_t slot = from + blockIdx.x * 256 + threadIdx.x;
    if (slot < D.P) { uint2 z; z.x = 0; z.y = 0; D.st2(SF_MA, slot, z); }
}

// one wave per pixel: lane l sums samples l, l + 64, ... in order, then a fixed xor butterfly — the order depends on
// spp only, never on which slot produced a sample
__global__ __launch_bounds__(256) void stream_reduce(StreamBuf B, DCamera cam, double* __restrict__ out) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= B.n_pix) return;
    const double* pp = B.samples + (size_t)i * B.spp * 3;
    double sx = 0, sy = 0, sz = 0;
    for (uint32_t sidx = (uint32_t)lane; sidx < B.spp; sidx += 64) { sx += pp[sidx * 3]; sy += pp[sidx * 3 + 1]; sz += pp[sidx * 3 + 2]; }
    for (int m = 32; m >= 1; m >>= 1) {
        sx += __hiloint2double(__shfl_xor(__double2hiint(sx), m, 64), __shfl_xor(__double2loint(sx), m, 64));
        sy += __hiloint2double(__shfl_xor(__double2hiint(sy), m, 64), __shfl_xor(__double2loint(sy), m, 64));
        sz += __hiloint2double(__shfl_xor(__double2hiint(sz), m, 64), __shfl_xor(__double2loint(sz), m, 64));
    }
    if (lane == 0) {
        const uint32_t pk = B.pixels[i];
        const int px = (int)(pk & 0xFFFFu), py = (int)(pk >> 16);
        const double scale = 1.0 / cam.spp;  // camera.hpp:437,531
        double* o = out + ((size_t)py * cam.W + px) * 3;
        o[0] = sx * scale; o[1] = sy * scale; o[2] = sz * scale;
    }
}

// per-block counters of the split passes -> the context's counter words (samples, segments, hits, draws)
__global__ __launch_bounds__(256) void stream_sum_counters(const unsigned long long* __restrict__ cpart, size_t n_blocks, unsigned long long* __restrict__ gctr) {
    __shared__ unsigned long long s[4];
    if (threadIdx.x < 4) s[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long a[4] = {0, 0, 0, 0};
    for (size_t b = threadIdx.x; b < n_blocks; b += 256) for (int k = 0; k < 4; k++) a[k] += cpart[b * 4 + k];
    for (int k = 0; k < 4; k++) if (a[k]) atomicAdd(&s[k], a[k]);
    __syncthreads();
    if (threadIdx.x == 0) { gctr[0] += s[0]; gctr[1] += s[1]; gctr[7] += s[2]; gctr[8] += s[3]; }
}

// the split frames: the same order-fixed sum, every sample to the frame its class names
__global__ __launch_bounds__(256) void stream_reduce_split(StreamBuf B, DCamera cam, double* __restrict__ out_reflection, double* __restrict__ out_refraction) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= B.n_pix) return;
    const double* pp = B.samples + (size_t)i * B.spp * 3;
    const unsigned char* cc = B.cls + (size_t)i * B.spp;
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t sidx = (uint32_t)lane; sidx < B.spp; sidx += 64) {
        const unsigned char c = cc[sidx];
        if (c == 1) { a[0] += pp[sidx * 3]; a[1] += pp[sidx * 3 + 1]; a[2] += pp[sidx * 3 + 2]; }
        else if (c == 2) { a[3] += pp[sidx * 3]; a[4] += pp[sidx * 3 + 1]; a[5] += pp[sidx * 3 + 2]; }
    }
    for (int m = 32; m >= 1; m >>= 1)
        for (int k = 0; k < 6; k++) a[k] += __hiloint2double(__shfl_xor(__double2hiint(a[k]), m, 64), __shfl_xor(__double2loint(a[k]), m, 64));
    if (lane == 0) {
        const uint32_t pk = B.pixels[i];
        const int px = (int)(pk & 0xFFFFu), py = (int)(pk >> 16);
        const double scale = 1.0 / cam.spp;  // camera.hpp:532-533
        const size_t o = ((size_t)py * cam.W + px) * 3;
        if (out_reflection) { out_reflection[o] = a[0] * scale; out_reflection[o + 1] = a[1] * scale; out_reflection[o + 2] = a[2] * scale; }
        if (out_refraction) { out_refraction[o] = a[3] * scale; out_refraction[o + 1] = a[4] * scale; out_refraction[o + 2] = a[5] * scale; }
    }
}

// ---- known-answer path: a batch of caller-supplied rays through EXTEND (zr_trace) ------------------------------------
// slot k = ray k, RNG key as in trace_rays (zr_kernels.hip) so that a medium's draw is the same on both engines
__global__ __launch_bounds__(256) void stream_load_rays(StreamBuf B, const double* __restrict__ rays, uint32_t n, uint64_t seed, uint64_t pixel,
                                                         uint32_t bounce) {
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= B.P) return;
    uint2 ma; ma.x = 0; ma.y = 0;
    if (slot < n) {
        B.st3(SF_RAY, slot, ld3(rays + (size_t)slot * 6)); B.st3(SF_RAY + 3, slot, ld3(rays + (size_t)slot * 6 + 3));
        B.st(SF_KEY, slot, __longlong_as_double((long long)zr_stream_key(seed, pixel, (uint64_t)slot)));
        ma.y = (bounce & 0xFFu) | F_ACTIVE;
    }
    B.st2(SF_MA, slot, ma);
}

__global__ __launch_bounds__(256) void stream_hits_out(DScene sc, StreamBuf B, uint32_t n, zr_hit* __restrict__ out) {
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= n) return;
    const uint2 ki = B.ld2(SF_HIT_KI, slot);
    zr_hit o;
    if (ki.x != 0xFFFFFFFFu) {
        Ray r; r.o = B.ld3(SF_RAY, slot); r.d = B.ld3(SF_RAY + 3, slot);
        Rec rec;
        object_rec(sc, ki.x, ki.y, r, B.ld(SF_HIT_T, slot), rec, true);
        o.p[0] = rec.p.x; o.p[1] = rec.p.y; o.p[2] = rec.p.z;
        o.normal[0] = rec.n.x; o.normal[1] = rec.n.y; o.normal[2] = rec.n.z;
        o.tangent[0] = rec.tan.x; o.tangent[1] = rec.tan.y; o.tangent[2] = rec.tan.z;
        o.bitangent[0] = rec.bit.x; o.bitangent[1] = rec.bit.y; o.bitangent[2] = rec.bit.z;
        o.t = rec.t; o.u = rec.u; o.v = rec.v; o.mat = rec.mat; o.front_face = rec.front ? 1u : 0u;
    } else {
        for (int c = 0; c < 3; c++) { o.p[c] = 0; o.normal[c] = 0; o.tangent[c] = 0; o.bitangent[c] = 0; }
        o.t = 0; o.u = 0; o.v = 0; o.mat = 0xFFFFFFFFu; o.front_face = 0;
    }
    out[slot] = o;
}

// ---- host-side launch helpers -----------------------------------------------------------------------------------
size_t stream_ctl_words() { return 16 + 32 * ST_SHARDS; }
// spill slab of the EXTEND stack: `stack_demand` = worst-case entries of the committed tree (zr_scene_stats), ST_LDS_STACK of them live in LDS
uint32_t stream_overflow_levels(uint32_t stack_demand) { return stack_demand > ST_LDS_STACK ? stack_demand - ST_LDS_STACK : 1u; }
size_t stream_overflow_bytes(int blocks, uint32_t levels) { return (size_t)blocks * levels * 64 * sizeof(SEntry); }

// number of EXTEND waves that are resident at once (the persistent grid), a multiple of the workgroup's ST_EXT_GROUP
int stream_extend_blocks() {
    int dev = 0, cus = 256, per_cu = 2;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    }
    int lean = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stream_extend<false, 2>, 64 * ST_EXT_GROUP, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&lean, stream_extend<false, 0>, 64 * ST_EXT_GROUP, 0) == hipSuccess && lean > per_cu) per_cu = lean;
    if (std::getenv("ZR_COMMIT_STATS")) std::fprintf(stderr, "[zr] EXTEND workgroups per CU: %d (lean build %d) x %d waves\n", per_cu, lean, ST_EXT_GROUP);
    return cus * per_cu * ST_EXT_GROUP;
}

// layout of the slot pool inside one allocation; returns bytes needed
size_t stream_pool_bytes(uint32_t P) { return ((size_t)(P + 63) / 64) * SF_N * 64 * sizeof(double); }

static StreamBuf make_buf(void* pool, uint32_t P, uint32_t spp, uint32_t n_units, uint32_t n_pix, const uint32_t* pixels, double* samples,
                          unsigned int* ctl, unsigned int* uctl, uint32_t unit0, uint32_t unit_base) {
    StreamBuf B;
    B.pool = (double*)pool; B.pixels = pixels; B.samples = samples; B.ctl = ctl; B.uctl = uctl;
    B.P = P; B.spp = spp; B.n_units = n_units; B.n_pix = n_pix; B.unit0 = unit0; B.unit_base = unit_base; B.unit_chunk = 0; B.shard_k0 = 0;
    B.kend = nullptr; B.cls = nullptr; B.cpart = nullptr;
    return B;
}

template <bool COUNT>
static void launch_extend(const DScene& sc, const StreamBuf& B, void* overflow, uint32_t ovf_levels, int blocks, unsigned long long* gctr, int level, hipStream_t st) {
    // `blocks` counts waves (at most stream_extend_blocks(), a multiple of ST_EXT_GROUP: the overflow slabs are sized for that many)
    const dim3 grid((blocks + ST_EXT_GROUP - 1) / ST_EXT_GROUP), group(64 * ST_EXT_GROUP);
    if (level >= 3) hipLaunchKernelGGL((stream_extend<COUNT, 3>), grid, group, 0, st, sc, B, (SEntry*)overflow, ovf_levels, gctr);
    else if (level == 2) hipLaunchKernelGGL((stream_extend<COUNT, 2>), grid, group, 0, st, sc, B, (SEntry*)overflow, ovf_levels, gctr);
    else if (level == 1) hipLaunchKernelGGL((stream_extend<COUNT, 1>), grid, group, 0, st, sc, B, (SEntry*)overflow, ovf_levels, gctr);
    else hipLaunchKernelGGL((stream_extend<COUNT, 0>), grid, group, 0, st, sc, B, (SEntry*)overflow, ovf_levels, gctr);
}

// The slot pool can be split into K sub-pools that run a fraction of a round apart on K HIP streams, so that one
// pool's EXTEND overlaps another's SHADE.  Measured on cfg3: K = 2 gains 1-3 % on a whole frame and 6 % on a rank's
// 1/8 shard, K >= 3 loses (the stages are throughput-bound, co-running launches only stretch each other).
// streams[0] is the caller's stream, the others are internal.
hipError_t stream_render(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, void* pool, uint32_t P, uint32_t spp,
                         uint32_t n_pix, const uint32_t* d_pixels, double* d_samples, unsigned int* d_ctl, void* d_overflow, uint32_t ovf_levels, int extend_blocks,
                         double* out, unsigned long long* gctr, bool count, hipStream_t* streams, int n_pools, hipEvent_t ev, StreamTimer* timer,
                         unsigned int* h_active, volatile const uint8_t* keep_going, int* rounds_out, int generic, int mode, void* d_kend, void* d_cls,
                         double* out2, unsigned long long* d_cpart, StreamProgress* progress, void* drain_pool, uint32_t drain_slots, uint32_t unit_chunk) {
    const uint32_t n_units = n_pix * spp;
    const size_t W = stream_ctl_words();
    int K = n_pools < 1 ? 1 : (n_pools > ST_MAX_POOLS ? ST_MAX_POOLS : n_pools);
    while (K > 1 && P / K < 64u * 1024u) K--;
    // affine hand-out (st_unit_of): a slot's shard is (slot / 256) % ST_SHARDS counted over ALL sub-pools, and the SHADE block that
    // serves it must agree (blockIdx % ST_SHARDS inside its sub-pool): pool and sub-pools are whole rounds of ST_SHARDS blocks
    const uint32_t pool_round = unit_chunk ? ST_SHARDS * 256u : 64u;
    if (unit_chunk && P % pool_round != 0) return hipErrorInvalidValue;
    while (K > 1 && P / K < pool_round) K--;
    StreamBuf Q[ST_MAX_POOLS];
    void* ov[ST_MAX_POOLS];
    unsigned int* uctl = d_ctl + (size_t)ST_MAX_POOLS * W;
    {
        uint32_t first = 0;
        unsigned char* base = (unsigned char*)pool;
        for (int k = 0; k < K; k++) {
            uint32_t Pk = k == K - 1 ? P - first : (P / K + pool_round - 1) / pool_round * pool_round;
            Q[k] = make_buf(base, Pk, spp, n_units, n_pix, d_pixels, d_samples, d_ctl + (size_t)k * W, uctl, first, P);
            Q[k].unit_chunk = unit_chunk; Q[k].shard_k0 = unit_chunk ? P / ST_SHARDS : 0u;
            Q[k].kend = (uint2*)d_kend; Q[k].cls = (unsigned char*)d_cls;
            Q[k].cpart = d_cpart ? d_cpart + ((size_t)first / 256 + (size_t)k) * 4 : nullptr;
            ov[k] = (unsigned char*)d_overflow + (size_t)k * stream_overflow_bytes(extend_blocks, ovf_levels);
            base += stream_pool_bytes(Pk);
            first += Pk;
        }
    }
    hipError_t e;
    hipStream_t stream = streams[0];
    if ((e = hipMemsetAsync(d_ctl, 0, (ST_MAX_POOLS + 1) * W * sizeof(unsigned int), stream)) != hipSuccess) return e;
    const size_t cpart_blocks = (size_t)P / 256 + ST_MAX_POOLS + 1;
    if (d_cpart && (e = hipMemsetAsync(d_cpart, 0, cpart_blocks * 4 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
    auto init = [&](const StreamBuf& B, hipStream_t st) {
        if (timer) timer->begin(st, 0);
        if (count) hipLaunchKernelGGL(stream_init<true>, dim3((B.P + 255) / 256), dim3(256), 0, st, B, cam, seed, gctr);
        else hipLaunchKernelGGL(stream_init<false>, dim3((B.P + 255) / 256), dim3(256), 0, st, B, cam, seed, gctr);
        if (timer) timer->end(st, 0);
    };
    auto extend = [&](const StreamBuf& B, void* o, hipStream_t st) {
        const int eb = (int)((B.P + 63) / 64 < (uint32_t)extend_blocks ? (B.P + 63) / 64 : (uint32_t)extend_blocks);
        if (timer) timer->begin(st, 1);
        if (count) launch_extend<true>(sc, B, o, ovf_levels, eb, gctr, generic, st); else launch_extend<false>(sc, B, o, ovf_levels, eb, gctr, generic, st);
        if (timer) timer->end(st, 1);
    };
    // development aid: ZR_SHADE_LDS_PAD bytes of dynamic LDS per SHADE block lower the blocks a CU can hold (160 KB: 4 fit by registers; > 40 KB: 3, > 53 KB: 2)
    static const unsigned shade_pad = std::getenv("ZR_SHADE_LDS_PAD") ? (unsigned)std::atoi(std::getenv("ZR_SHADE_LDS_PAD")) : 0u;
    auto shade = [&](const StreamBuf& B, hipStream_t st) {
        if (timer) timer->begin(st, 2);
        const dim3 sg((B.P + 255) / 256), sb(256);
        if (sc.shade_lean && mode == 0) {   // the lean build: same arithmetic, fewer registers, 6 waves per SIMD instead of 4 (see lean_rec, zr_device.h)
            if (count) hipLaunchKernelGGL((stream_shade<true, 0, true>), sg, sb, 0, st, sc, cam, env, seed, B, gctr);
            else hipLaunchKernelGGL((stream_shade<false, 0, true>), sg, sb, shade_pad, st, sc, cam, env, seed, B, gctr);
            if (timer) timer->end(st, 2);
            return;
        }
        if (shade_pad && mode == 0 && !count) {
            hipLaunchKernelGGL((stream_shade<false, 0>), sg, sb, shade_pad, st, sc, cam, env, seed, B, gctr);
            if (timer) timer->end(st, 2);
            return;
        }
        if (mode == 1) hipLaunchKernelGGL((stream_shade<true, 1>), sg, sb, 0, st, sc, cam, env, seed, B, gctr);        // the split passes always count
        else if (mode == 2) hipLaunchKernelGGL((stream_shade<true, 2>), sg, sb, 0, st, sc, cam, env, seed, B, gctr);
        else if (count) hipLaunchKernelGGL((stream_shade<true, 0>), sg, sb, 0, st, sc, cam, env, seed, B, gctr);
        else hipLaunchKernelGGL((stream_shade<false, 0>), sg, sb, 0, st, sc, cam, env, seed, B, gctr);
        if (timer) timer->end(st, 2);
    };
    // start-up: pool k is initialised after the control words are cleared and starts once pool k-1 has a round in flight
    init(Q[0], stream);
    for (int k = 1; k < K; k++) {
        extend(Q[k - 1], ov[k - 1], streams[k - 1]);
        if ((e = hipEventRecord(ev, streams[k - 1])) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(streams[k], ev, 0)) != hipSuccess) return e;
        init(Q[k], streams[k]);
        shade(Q[k - 1], streams[k - 1]);
    }
    int rounds = K > 1 ? 1 : 0;
    int check_every = progress ? 2 : 8;
    bool cancelled = false, drained = false, capped = false;
    for (;;) {
        for (int r = 0; r < check_every; r++) {
            for (int k = K - 1; k >= 0; k--) { extend(Q[k], ov[k], streams[k]); shade(Q[k], streams[k]); }
            rounds++;
        }
        for (int k = 0; k < K && e == hipSuccess; k++)
            e = hipMemcpyAsync(h_active + (size_t)k * W, d_ctl + (size_t)k * W, W * sizeof(unsigned int), hipMemcpyDeviceToHost, streams[k]);
        if (e == hipSuccess)   // the work-unit counters: how many samples have been started
            e = hipMemcpyAsync(h_active + (size_t)ST_MAX_POOLS * W, uctl, W * sizeof(unsigned int), hipMemcpyDeviceToHost, streams[0]);
        for (int k = K - 1; k >= 0 && e == hipSuccess; k--) e = hipStreamSynchronize(streams[k]);
        if (e != hipSuccess) break;
        unsigned long long active = 0;
        for (int k = 0; k < K; k++) {
            if (h_active[(size_t)k * W + 2] != 0) capped = true;   // (looked at here, for every sub-pool: the drain below folds them into one)
            for (int sh = 0; sh < ST_SHARDS; sh++) active += h_active[(size_t)k * W + 16 + 32 * sh + 16];
        }
        if (progress && active != 0) {
            unsigned long long started = 0;   // the first P units are dealt at initialisation, the rest through the sharded counters
            if (!unit_chunk) { started = P; for (int sh = 0; sh < ST_SHARDS; sh++) started += h_active[(size_t)ST_MAX_POOLS * W + 32 * sh]; }
            else for (int sh = 0; sh < ST_SHARDS; sh++) {   // affine: of a shard's sequence, the units that exist (a shard's last chunks may lie beyond the frame)
                const unsigned long long k_sh = (unsigned long long)(P / ST_SHARDS) + h_active[(size_t)ST_MAX_POOLS * W + 32 * sh];
                const uint32_t pos = ((uint32_t)sh & 7u) * (ST_SHARDS / 8u) + ((uint32_t)sh >> 3);
                const unsigned long long full = k_sh / unit_chunk, part = k_sh % unit_chunk;
                for (unsigned long long q = 0; q <= full; q++) {
                    const unsigned long long c0 = (q * ST_SHARDS + pos) * unit_chunk, len = q < full ? unit_chunk : part;
                    if (c0 < n_units) started += c0 + len <= n_units ? len : n_units - c0;
                }
            }
            if (started > n_units) started = n_units;
            const double frac = started > active ? (double)(started - active) / (double)n_units : 0.0;
            bool reduced = false;
            if (mode == 0 && out && progress->wants_frame()) {   // samples[] is zero-initialised in this case (render_stream)
                hipLaunchKernelGGL(stream_reduce, dim3((n_pix + 3) / 4), dim3(256), 0, stream, Q[0], cam, out);
                if ((e = hipStreamSynchronize(stream)) != hipSuccess) break;
                reduced = true;
            }
            progress->report(frac, reduced);
        }
        if (active == 0) break;
        if (!drained && drain_pool && active * 16ull <= (unsigned long long)P && active <= (unsigned long long)drain_slots) {
            bool units_left = false;   // shard sh has handed out every k below its counter: unit_base + k * shards + sh
            for (int sh = 0; sh < ST_SHARDS && !units_left; sh++)
                if (st_unit_of(unit_chunk, P, (uint32_t)sh, (unsigned long long)(unit_chunk ? P / ST_SHARDS : 0u) + h_active[(size_t)ST_MAX_POOLS * W + 32 * sh]) < (unsigned long long)n_units) units_left = true;
            if (!units_left) {   // see stream_compact
                StreamBuf D = Q[0];
                D.pool = (double*)drain_pool; D.P = (uint32_t)((active + 255ull) / 256ull * 256ull); D.unit0 = 0;
                unsigned int* counter = d_ctl + 4;   // a control word of pool 0 nothing else uses; zero since the frame began
                for (int k = 0; k < K; k++) hipLaunchKernelGGL(stream_compact, dim3((Q[k].P + 255) / 256), dim3(256), 0, stream, Q[k], D, counter);
                if (D.P > (uint32_t)active) hipLaunchKernelGGL(stream_compact_pad, dim3((D.P - (uint32_t)active + 255) / 256), dim3(256), 0, stream, D, (uint32_t)active);
                if ((e = hipStreamSynchronize(stream)) != hipSuccess) break;
                Q[0] = D; K = 1; drained = true;
            }
        }
        check_every = active > P / 2 ? 8 : (active > P / 16 ? 4 : 2);
        if (progress && check_every > 2) check_every = 2;   // an interactive caller: report (and poll keep_going) every other round
        if (keep_going && *keep_going == 0) { cancelled = true; break; }
        if (rounds > (1 << 22)) { e = hipErrorLaunchFailure; break; }
    }
    if (e != hipSuccess) return e;
    if (capped) return hipErrorLaunchFailure;  // an EXTEND wave of some sub-pool hit its iteration cap: its rays were abandoned mid-walk
    const StreamBuf& A = Q[0];
    if (timer) timer->begin(stream, 3);
    if (d_cpart) hipLaunchKernelGGL(stream_sum_counters, dim3(1), dim3(256), 0, stream, d_cpart, cpart_blocks, gctr);
    if (mode == 2) hipLaunchKernelGGL(stream_reduce_split, dim3((n_pix + 3) / 4), dim3(256), 0, stream, A, cam, out, out2);
    else if (out) hipLaunchKernelGGL(stream_reduce, dim3((n_pix + 3) / 4), dim3(256), 0, stream, A, cam, out);
    if (timer) timer->end(stream, 3);
    if (rounds_out) *rounds_out = cancelled ? -rounds : rounds;
    return hipGetLastError();
}

// the fused small-scene render: one persistent launch over all units, then the pipeline's reduce.  `level`: 1 = no wrapped objects
// and only plain media, 2 = everything but placements
int fused_blocks() {
    int dev = 0, cus = 256, per_cu = 2;
    if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_render<2, false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    return cus * per_cu;
}
hipError_t fused_render_frame(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, uint32_t spp, uint32_t n_pix, const uint32_t* d_pixels, double* d_samples,
                              unsigned int* d_ctl, int blocks, double* out, unsigned long long* gctr, bool count, int level, hipStream_t stream, StreamTimer* timer,
                              const FusedObjs& fo, volatile const uint8_t* keep_going, StreamProgress* progress, int* parts_done) {
    const uint32_t n_units = n_pix * spp;
    const size_t W = stream_ctl_words();
    hipError_t e;
    FusedBuf B; B.pixels = d_pixels; B.samples = d_samples; B.uctl = d_ctl + (size_t)ST_MAX_POOLS * W; B.spp = spp; B.n_units = n_units;
    const unsigned long long n_chunks = ((unsigned long long)n_units + ST_FUSED_CHUNK - 1) / ST_FUSED_CHUNK;
    // one launch for the whole frame — or, for a caller that polls (render_flag, lines_rendered, the live preview of camera.hpp:548-552 /
    // main.cpp:1576), sixteen launches over consecutive parts of the unit range with the poll between them (samples[] was zeroed, so the
    // reduce of a partial frame is the mean of the samples finished so far, as in the pipeline)
    const int parts = (keep_going || progress) && n_chunks >= 64 ? 16 : 1;
    StreamBuf R = make_buf(nullptr, 0, spp, n_units, n_pix, d_pixels, d_samples, d_ctl, d_ctl, 0, 0);
    bool cancelled = false;
    int done = 0;
    for (int p = 0; p < parts && !cancelled; p++) {
        B.chunk_lo = (uint32_t)(n_chunks * (unsigned long long)p / (unsigned long long)parts);
        B.chunk_hi = (uint32_t)(n_chunks * (unsigned long long)(p + 1) / (unsigned long long)parts);
        if (B.chunk_hi == B.chunk_lo) continue;
        if ((e = hipMemsetAsync(d_ctl, 0, (ST_MAX_POOLS + 1) * W * sizeof(unsigned int), stream)) != hipSuccess) return e;
        const unsigned long long want_blocks = ((unsigned long long)(B.chunk_hi - B.chunk_lo) + 3) / 4;   // no more waves than chunks
        const dim3 grid((unsigned)(want_blocks < (unsigned long long)blocks ? (want_blocks ? want_blocks : 1) : blocks)), block(256);
        if (timer) timer->begin(stream, 1);
        if (level <= 1) { if (count) hipLaunchKernelGGL((fused_render<1, true>), grid, block, 0, stream, sc, cam, env, seed, B, gctr, fo); else hipLaunchKernelGGL((fused_render<1, false>), grid, block, 0, stream, sc, cam, env, seed, B, gctr, fo); }
        else { if (count) hipLaunchKernelGGL((fused_render<2, true>), grid, block, 0, stream, sc, cam, env, seed, B, gctr, fo); else hipLaunchKernelGGL((fused_render<2, false>), grid, block, 0, stream, sc, cam, env, seed, B, gctr, fo); }
        if (timer) timer->end(stream, 1);
        done = p + 1;
        if (parts > 1 && p + 1 < parts) {
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
            if (progress) {
                bool reduced = false;
                if (out && progress->wants_frame()) {
                    hipLaunchKernelGGL(stream_reduce, dim3((n_pix + 3) / 4), dim3(256), 0, stream, R, cam, out);
                    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
                    reduced = true;
                }
                progress->report((double)(p + 1) / parts, reduced);
            }
            if (keep_going && *keep_going == 0) cancelled = true;
        }
    }
    if (timer) timer->begin(stream, 3);
    if (out) hipLaunchKernelGGL(stream_reduce, dim3((n_pix + 3) / 4), dim3(256), 0, stream, R, cam, out);
    if (timer) timer->end(stream, 3);
    if (parts_done) *parts_done = cancelled ? -done : done;
    return hipGetLastError();
}

// closest hits of n rays in [0.001, inf) through the EXTEND kernel; `pool` holds stream_pool_bytes(round_up(n, 64)) bytes
hipError_t stream_trace(const DScene& sc, const double* d_rays, uint32_t n, uint64_t seed, uint64_t pixel, uint32_t bounce, zr_hit* d_out,
                        void* pool, unsigned int* d_ctl, void* d_overflow, uint32_t ovf_levels, int extend_blocks, unsigned long long* gctr, int generic,
                        hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t P = (n + 63u) / 64u * 64u;
    const size_t W = stream_ctl_words();
    StreamBuf B = make_buf(pool, P, 1, n, n, nullptr, nullptr, d_ctl, d_ctl + (size_t)ST_MAX_POOLS * W, 0, P);
    hipError_t e;
    if ((e = hipMemsetAsync(d_ctl, 0, (ST_MAX_POOLS + 1) * W * sizeof(unsigned int), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(stream_load_rays, dim3((P + 255) / 256), dim3(256), 0, stream, B, d_rays, n, seed, pixel, bounce);
    const int eb = (int)(P / 64 < (uint32_t)extend_blocks ? P / 64 : (uint32_t)extend_blocks);
    launch_extend<false>(sc, B, d_overflow, ovf_levels, eb, gctr, generic, stream);
    hipLaunchKernelGGL(stream_hits_out, dim3((n + 255) / 256), dim3(256), 0, stream, sc, B, n, d_out);
    return hipGetLastError();
}

}  // namespace zr
