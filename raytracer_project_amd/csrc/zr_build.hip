// zr_build.hip — the BVH build on the device (row f-3 of SURVEY.md §8): see zr_build.h for what it replaces and produces.
//
// Pipeline (every step a kernel over HBM-resident arrays; n = world-list entries):
//   1. boxes      one thread per object: its box the way the reference's constructors compute it (sphere.hpp:12-14,
//                 triangle.hpp:84-101, cube.hpp:34-41, the wrappers' constructors), rounded OUTWARDS to float; scene bounds by
//                 wave reduction + one atomic per wave
//   2. keys       63-bit Morton code of the box centre on the scene's grid; rocPRIM radix sort of (key, object)
//   3. PLOC       clusters in Morton order; per iteration: every cluster finds the neighbour within `radius` positions whose union
//                 with it has the smallest surface area (boxes of a 256-cluster tile + halo staged through LDS), mutual nearest
//                 neighbours merge into a new node, the survivors are compacted IN ORDER (block counts -> one-block scan ->
//                 scatter), so the tree is a pure function of the input.  A merged node knows at once what its subtree holds
//                 (count, kind, SAH cost), so the leaf decision — collapse a subtree of <= max_leaf primitives of one kind when
//                 testing them costs less than walking it — is taken at the merge: no bottom-up pass, no inter-workgroup hand-off.
//   4. order      every final leaf climbs to the root once: its depth-first position (sum of the left siblings' counts) and the
//                 tree's depth; primitives laid out in depth-first order, per kind (one scan per kind present)
//   5. 4-wide     nodes are created in iteration batches, a parent always after its children: walking the batches in REVERSE is a
//                 top-down pass without a queue.  A node that is a quad root opens its largest inner children while the 8-bit grid
//                 of the wider node keeps every box within `open_ratio` of its true area (Flattener::plan_quad's rule), writes
//                 its 64-byte record and names its inner children quad roots.  Forward over the batches: exact stack demand.
//   6. pairs      one 64-byte sibling-pair record per inner node (pair-BVH walk of the megakernel / zr_trace / AOV kernels)
//   7. emit       one thread per primitive: its record(s) in leaf order (bare, baked, placed: the host's put_* rules)
// Media and wrapped objects (a handful per scene, each dragging inner primitives behind it) are listed for the host to finish.
//
// Quantisation is exact without long double: origin + q * scale (float + small integer x power of two) is compared with a box
// plane through an error-free TwoSum in double, so "the grid box CONTAINS the true box" is decided on real numbers, as the host's
// long-double check does.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "zr_build.h"
#include "zr_bvh.h"

namespace zr {
namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int MAX_R = 32;
constexpr uint32_t M_KIND = 7u, M_UNIFORM = 8u, M_LEAF = 16u;   // node meta: kind | uniform | leaf | small count << 8

struct PBox { float lo[3]; uint32_t kind; float hi[3]; uint32_t pad; };
struct Cl { float lo[3]; uint32_t node; float hi[3]; uint32_t pad; };
struct BNode { float lo[3]; uint32_t left; float hi[3]; uint32_t right; };   // primitive: left = object, right = NONE
static_assert(sizeof(PBox) == 32 && sizeof(Cl) == 32 && sizeof(BNode) == 32, "32-byte records: two 16-byte accesses");

struct Params {   // BuildParams by value in kernel arguments
    float ct; float ck[8]; int max_leaf; int cap[8]; float open_ratio; int radius;
};

__device__ __forceinline__ float f_down(double x) { return nextafterf(__double2float_rd(x), -__builtin_huge_valf()); }
__device__ __forceinline__ float f_up(double x) { return nextafterf(__double2float_ru(x), __builtin_huge_valf()); }
__device__ __forceinline__ float half_area(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// ---- 1. boxes ---------------------------------------------------------------------------------------------------------
struct DBox { double lo[3], hi[3]; };

__device__ DBox prim_box(const BuildSceneIn& in, uint32_t type, uint32_t idx);
__device__ DBox apply_op_box(const zr_xform_op& op, const DBox& b) {
    DBox o;
    if (op.kind == ZR_OP_TRANSLATE) { for (int k = 0; k < 3; k++) { o.lo[k] = b.lo[k] + op.a[k]; o.hi[k] = b.hi[k] + op.a[k]; } return o; }   // translate.hpp:12
    if (op.kind == ZR_OP_SCALE) {   // scale.hpp:11-17
        for (int k = 0; k < 3; k++) { const double a0 = b.lo[k] * op.a[k], a1 = b.hi[k] * op.a[k]; o.lo[k] = fmin(a0, a1); o.hi[k] = fmax(a0, a1); }
        return o;
    }
    if (op.kind == ZR_OP_MATERIAL) return b;
    // rotate_*.hpp constructors: the box of the eight rotated corners (rotate_y: of the TRUE geometry).  A rotation
    // about one axis turns the rectangle [a0, a1] x [b0, b1] of the other two; its four corners, with the constructors' arithmetic
    // (scalars only: no indexed local arrays — the first version indexed a local double[3] inside the corner loop and
    // came back from the gfx950 compiler with the unrotated coordinate overwritten, found by ZR_BUILD_CHECK)
    const double sn = op.a[0], co = op.a[1];
    const int ia = op.kind == ZR_OP_ROTATE_X ? 1 : 0, ib = op.kind == ZR_OP_ROTATE_Z ? 1 : 2, ic = 3 - ia - ib;   // (a, b) plane, c = the axis
    const double a0 = ia == 0 ? b.lo[0] : b.lo[1], a1 = ia == 0 ? b.hi[0] : b.hi[1];
    const double b0 = ib == 1 ? b.lo[1] : b.lo[2], b1 = ib == 1 ? b.hi[1] : b.hi[2];
    const double c0 = ic == 0 ? b.lo[0] : (ic == 1 ? b.lo[1] : b.lo[2]), c1 = ic == 0 ? b.hi[0] : (ic == 1 ? b.hi[1] : b.hi[2]);
    // rotate_y: x' = co x - sn z, z' = sn x + co z   (a = x, b = z)
    // rotate_x: y' = co y - sn z, z' = sn y + co z   (a = y, b = z)
    // rotate_z: x' = co x - sn y, y' = sn x + co y   (a = x, b = y)
    const double u00 = co * a0 - sn * b0, u10 = co * a1 - sn * b0, u01 = co * a0 - sn * b1, u11 = co * a1 - sn * b1;
    const double v00 = sn * a0 + co * b0, v10 = sn * a1 + co * b0, v01 = sn * a0 + co * b1, v11 = sn * a1 + co * b1;
    const double ulo = fmin(fmin(u00, u10), fmin(u01, u11)), uhi = fmax(fmax(u00, u10), fmax(u01, u11));
    const double vlo = fmin(fmin(v00, v10), fmin(v01, v11)), vhi = fmax(fmax(v00, v10), fmax(v01, v11));
    if (op.kind == ZR_OP_ROTATE_Y) { o.lo[0] = ulo; o.hi[0] = uhi; o.lo[1] = c0; o.hi[1] = c1; o.lo[2] = vlo; o.hi[2] = vhi; }
    else if (op.kind == ZR_OP_ROTATE_X) { o.lo[0] = c0; o.hi[0] = c1; o.lo[1] = ulo; o.hi[1] = uhi; o.lo[2] = vlo; o.hi[2] = vhi; }
    else { o.lo[0] = ulo; o.hi[0] = uhi; o.lo[1] = vlo; o.hi[1] = vhi; o.lo[2] = c0; o.hi[2] = c1; }
    return o;
}
__device__ DBox chain_box(const BuildSceneIn& in, uint32_t type, uint32_t idx, uint32_t cf, uint32_t cn) {
    DBox b = prim_box(in, type, idx);
    for (int k = (int)cn - 1; k >= 0; k--) b = apply_op_box(in.ops[cf + k], b);   // innermost wrapper first
    return b;
}
__device__ DBox prim_box(const BuildSceneIn& in, uint32_t type, uint32_t idx) {
    DBox b;
    if (type == ZR_PRIM_GROUP) { for (int k = 0; k < 3; k++) { b.lo[k] = in.group_box[(size_t)idx * 6 + k]; b.hi[k] = in.group_box[(size_t)idx * 6 + 3 + k]; } return b; }
    if (type == ZR_PRIM_SPHERE) {   // sphere.hpp:12-14 (raw radius argument)
        const double* q = in.spheres + (size_t)idx * 4;
        for (int k = 0; k < 3; k++) { b.lo[k] = fmin(q[k] - q[3], q[k] + q[3]); b.hi[k] = fmax(q[k] - q[3], q[k] + q[3]); }
    } else if (type == ZR_PRIM_TRIANGLE) {   // triangle.hpp:84-101
        const double* v = in.tri_v + (size_t)idx * 9;
        for (int k = 0; k < 3; k++) {
            b.lo[k] = fmin(v[k], fmin(v[3 + k], v[6 + k]));
            b.hi[k] = fmax(v[k], fmax(v[3 + k], v[6 + k]));
            if (b.hi[k] - b.lo[k] < 0.0001) { b.lo[k] -= 0.0001; b.hi[k] += 0.0001; }
        }
    } else if (type == ZR_PRIM_CUBE) {   // cube.hpp:34-41
        const double* q = in.cubes + (size_t)idx * 12;
        for (int k = 0; k < 3; k++) { b.lo[k] = q[6 + k] - 0.00005; b.hi[k] = q[9 + k] + 0.00005; }
    } else {   // constant_medium.hpp:79-81: the boundary's box (a sphere or a cube under the medium's own chain)
        const zr_medium m = in.media[idx];
        DBox bb;
        if (m.boundary_type == ZR_PRIM_SPHERE) {
            const double* q = in.spheres + (size_t)m.boundary_index * 4;
            for (int k = 0; k < 3; k++) { bb.lo[k] = fmin(q[k] - q[3], q[k] + q[3]); bb.hi[k] = fmax(q[k] - q[3], q[k] + q[3]); }
        } else {
            const double* q = in.cubes + (size_t)m.boundary_index * 12;
            for (int k = 0; k < 3; k++) { bb.lo[k] = q[6 + k] - 0.00005; bb.hi[k] = q[9 + k] + 0.00005; }
        }
        for (int k = (int)m.chain_count - 1; k >= 0; k--) bb = apply_op_box(in.ops[m.chain_first + k], bb);
        b = bb;
    }
    return b;
}

// order-preserving map float <-> uint for atomic min / max
__device__ __forceinline__ uint32_t f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// bounds[0..2] = min of the box centres, [3..5] = max (ordered uints); bounds[6] = 1 when a box is not usable
__global__ __launch_bounds__(256) void k_boxes(BuildSceneIn in, const zr_object* __restrict__ objs, const uint8_t* __restrict__ code, uint32_t first_triangle,
                                               uint32_t n, PBox* __restrict__ pbox, uint32_t* __restrict__ bounds) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float c[3] = {0, 0, 0};
    bool ok = true;
    const bool live = i < n;
    if (live) {
        DBox b; uint32_t kind;
        if (objs) { const zr_object o = objs[i]; b = chain_box(in, o.type, o.index, o.chain_first, o.chain_count); kind = code[i] & 7u; }
        else { b = prim_box(in, ZR_PRIM_TRIANGLE, first_triangle + i); kind = ZR_PRIM_TRIANGLE; }
        PBox p;
        for (int k = 0; k < 3; k++) {
            p.lo[k] = f_down(b.lo[k]); p.hi[k] = f_up(b.hi[k]);
            if (!(fabs(b.lo[k]) < 1e18) || !(fabs(b.hi[k]) < 1e18)) ok = false;   // (also catches NaN and infinities)
            c[k] = 0.5f * p.lo[k] + 0.5f * p.hi[k];
        }
        p.kind = kind; p.pad = 0;
        pbox[i] = p;
    }
    // wave reduction of the centre bounds, one atomic per wave and word
    for (int k = 0; k < 3; k++) {
        float mn = live ? c[k] : __builtin_huge_valf(), mx = live ? c[k] : -__builtin_huge_valf();
        for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, __shfl_xor(mn, m, 64)); mx = fmaxf(mx, __shfl_xor(mx, m, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bounds[k], f2o(mn)); atomicMax(&bounds[3 + k], f2o(mx)); }
    }
    if (__ballot(live && !ok) != 0ull && (threadIdx.x & 63) == 0) atomicExch(&bounds[6], 1u);
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {   // 21 bits -> every third bit
    x &= 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ __launch_bounds__(256) void k_keys(const PBox* __restrict__ pbox, uint32_t n, const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys,
                                              uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const PBox p = pbox[i];
    uint64_t key = 0;
    for (int k = 0; k < 3; k++) {
        const float lo = o2f(bounds[k]), hi = o2f(bounds[3 + k]);
        const float ext = hi - lo;
        const float c = 0.5f * p.lo[k] + 0.5f * p.hi[k];
        float u = ext > 0 ? (c - lo) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        uint32_t q = (uint32_t)(u * 2097151.0f);
        if (q > 2097151u) q = 2097151u;
        key |= spread21(q) << (2 - k);
    }
    keys[i] = key; vals[i] = i;
}

// ---- 3. PLOC ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t meta_of_prim(uint32_t kind) { return (kind & M_KIND) | M_UNIFORM | M_LEAF | (1u << 8); }

__global__ __launch_bounds__(256) void k_init_clusters(const PBox* __restrict__ pbox, const uint32_t* __restrict__ sorted, uint32_t n, Params prm,
                                                       Cl* __restrict__ cl, BNode* __restrict__ bn, uint32_t* __restrict__ ncount, uint32_t* __restrict__ nmeta,
                                                       float* __restrict__ ncost) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t obj = sorted[i];
    const PBox p = pbox[obj];
    Cl c; BNode b;
    for (int k = 0; k < 3; k++) { c.lo[k] = p.lo[k]; c.hi[k] = p.hi[k]; b.lo[k] = p.lo[k]; b.hi[k] = p.hi[k]; }
    c.node = i; c.pad = 0; b.left = obj; b.right = NONE;
    cl[i] = c; bn[i] = b;
    ncount[i] = 1; nmeta[i] = meta_of_prim(p.kind);
    ncost[i] = half_area(p.lo, p.hi) * prm.ck[p.kind & 7u];
}

// per tile of 256 clusters: nearest neighbours (for the tile and a halo of `r` on either side, so that mutuality is decided
// inside the block), what every cluster of the tile does — bit 0: keeps a slot in the next array, bit 1: creates a node — and the
// tile's totals.  Pairs are ranked by (merged area, distance in the order, parity of the lower index, lower index): a strict order
// on unordered pairs, so the best pair of any window is mutual (progress), and runs of identical boxes pair up (2k, 2k + 1)
// instead of all pointing at one cluster.
__global__ __launch_bounds__(256) void k_ploc_nn(const Cl* __restrict__ cin, uint32_t n_cur, int r, uint32_t* __restrict__ nn_out, uint8_t* __restrict__ act_out,
                                                 uint32_t* __restrict__ blk_keep, uint32_t* __restrict__ blk_new) {
    __shared__ float s_box[6][256 + 4 * MAX_R];
    __shared__ uint32_t s_nn[256 + 2 * MAX_R];
    __shared__ uint32_t s_cnt[2];
    const long base = (long)blockIdx.x * 256;
    if (base >= (long)n_cur) return;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    const int span = 256 + 4 * r;
    for (int t = threadIdx.x; t < span; t += 256) {
        const long g = base - 2 * r + t;
        if (g >= 0 && g < (long)n_cur) {
            const Cl c = cin[g];
            s_box[0][t] = c.lo[0]; s_box[1][t] = c.lo[1]; s_box[2][t] = c.lo[2];
            s_box[3][t] = c.hi[0]; s_box[4][t] = c.hi[1]; s_box[5][t] = c.hi[2];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 256 + 2 * r; e += 256) {
        const long g = base - r + e;
        uint32_t best_j = NONE;
        if (g >= 0 && g < (long)n_cur) {
            const int tg = e + r;   // position of g in s_box
            const float lx = s_box[0][tg], ly = s_box[1][tg], lz = s_box[2][tg], hx = s_box[3][tg], hy = s_box[4][tg], hz = s_box[5][tg];
            float best_c = __builtin_huge_valf(); int best_ad = 0x7FFFFFFF; uint32_t best_par = 2, best_mn = NONE;
            for (int d = -r; d <= r; d++) {
                const long j = g + d;
                if (d == 0 || j < 0 || j >= (long)n_cur) continue;
                const int tj = tg + d;
                const float dx = fmaxf(hx, s_box[3][tj]) - fminf(lx, s_box[0][tj]);
                const float dy = fmaxf(hy, s_box[4][tj]) - fminf(ly, s_box[1][tj]);
                const float dz = fmaxf(hz, s_box[5][tj]) - fminf(lz, s_box[2][tj]);
                const float cst = dx * dy + dy * dz + dz * dx;
                const int ad = d < 0 ? -d : d;
                const uint32_t mn = (uint32_t)(d < 0 ? j : g), par = mn & 1u;
                const bool better = cst < best_c || (cst == best_c && (ad < best_ad || (ad == best_ad && (par < best_par || (par == best_par && mn < best_mn)))));
                if (better || best_j == NONE) { best_c = cst; best_ad = ad; best_par = par; best_mn = mn; best_j = (uint32_t)j; }
            }
        }
        s_nn[e] = best_j;
    }
    __syncthreads();
    const long i = base + threadIdx.x;
    uint32_t keep = 0, mk = 0;
    if (i < (long)n_cur) {
        const uint32_t j = s_nn[threadIdx.x + r];
        bool mutual = false;
        if (j != NONE) mutual = s_nn[(long)j - (base - r)] == (uint32_t)i;
        if (mutual && (uint32_t)i < j) { keep = 1; mk = 1; }
        else if (!mutual) keep = 1;
        nn_out[i] = j;
        act_out[i] = (uint8_t)(keep | (mk << 1));
    }
    const unsigned long long bk = __ballot(keep != 0), bm = __ballot(mk != 0);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt[0], (uint32_t)__popcll(bk)); atomicAdd(&s_cnt[1], (uint32_t)__popcll(bm)); }
    __syncthreads();
    if (threadIdx.x == 0) { blk_keep[blockIdx.x] = s_cnt[0]; blk_new[blockIdx.x] = s_cnt[1]; }
}

// exclusive scan of the tiles' totals (one block; nb is a few thousand); state[0] = survivors, state[1] = new nodes
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint32_t* __restrict__ blk_keep, const uint32_t* __restrict__ blk_new, uint32_t nb,
                                                     uint32_t* __restrict__ keep_off, uint32_t* __restrict__ new_off, uint32_t* __restrict__ state) {
    __shared__ uint32_t s_a[1024], s_b[1024];
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    uint32_t a = 0, b = 0;
    for (uint32_t k = lo; k < hi; k++) { a += blk_keep[k]; b += blk_new[k]; }
    s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        uint32_t va = 0, vb = 0;
        if ((int)threadIdx.x >= off) { va = s_a[threadIdx.x - off]; vb = s_b[threadIdx.x - off]; }
        __syncthreads();
        s_a[threadIdx.x] += va; s_b[threadIdx.x] += vb;
        __syncthreads();
    }
    uint32_t ra = s_a[threadIdx.x] - a, rb = s_b[threadIdx.x] - b;   // exclusive
    for (uint32_t k = lo; k < hi; k++) { keep_off[k] = ra; new_off[k] = rb; ra += blk_keep[k]; rb += blk_new[k]; }
    if (threadIdx.x == 1023) { state[0] = s_a[1023]; state[1] = s_b[1023]; }
}

// what a merged node's subtree holds, and whether it is a leaf
__device__ __forceinline__ void merge_meta(const Params& prm, uint32_t ma, uint32_t mb, uint32_t ca, uint32_t cb, float cost_a, float cost_b, float area,
                                           uint32_t& count, uint32_t& meta, float& cost) {
    count = ca + cb;
    const uint32_t kind = ma & M_KIND;
    const bool uniform = (ma & M_UNIFORM) && (mb & M_UNIFORM) && kind == (mb & M_KIND);
    const bool both_leaf = (ma & M_LEAF) && (mb & M_LEAF);
    int cap = prm.cap[kind]; if (cap <= 0 || cap > prm.max_leaf) cap = prm.max_leaf;
    const float c_split = area * prm.ct + cost_a + cost_b;
    const float c_leaf = area * prm.ck[kind] * (float)count;
    const bool leaf = both_leaf && uniform && (int)count <= cap && c_leaf <= c_split;
    cost = leaf ? c_leaf : c_split;
    const uint32_t small = count > 255u ? 255u : count;
    meta = (uniform ? (kind | M_UNIFORM) : 0u) | (leaf ? M_LEAF : 0u) | (small << 8);
}

__global__ __launch_bounds__(256) void k_ploc_write(const Cl* __restrict__ cin, uint32_t n_cur, const uint32_t* __restrict__ nn, const uint8_t* __restrict__ act,
                                                    const uint32_t* __restrict__ keep_off, const uint32_t* __restrict__ new_off, uint32_t node_base, Params prm,
                                                    Cl* __restrict__ cout, BNode* __restrict__ bn, uint32_t* __restrict__ parent, uint32_t* __restrict__ ncount,
                                                    uint32_t* __restrict__ nmeta, float* __restrict__ ncost) {
    __shared__ uint32_t s_w[2][4];
    const long base = (long)blockIdx.x * 256;
    if (base >= (long)n_cur) return;
    const long i = base + threadIdx.x;
    const uint32_t a = i < (long)n_cur ? act[i] : 0u;
    const bool keep = a & 1u, mk = a & 2u;
    const int w = threadIdx.x >> 6, wl = threadIdx.x & 63;
    const unsigned long long below = (1ull << wl) - 1ull;
    const unsigned long long bk = __ballot(keep), bm = __ballot(mk);
    if (wl == 0) { s_w[0][w] = (uint32_t)__popcll(bk); s_w[1][w] = (uint32_t)__popcll(bm); }
    __syncthreads();
    uint32_t pos = keep_off[blockIdx.x] + (uint32_t)__popcll(bk & below), nid = new_off[blockIdx.x] + (uint32_t)__popcll(bm & below);
    for (int k = 0; k < w; k++) { pos += s_w[0][k]; nid += s_w[1][k]; }
    if (!keep) return;
    Cl c = cin[i];
    if (mk) {
        const Cl d = cin[nn[i]];
        const uint32_t id = node_base + nid;
        BNode b;
        for (int k = 0; k < 3; k++) { b.lo[k] = fminf(c.lo[k], d.lo[k]); b.hi[k] = fmaxf(c.hi[k], d.hi[k]); }
        b.left = c.node; b.right = d.node;
        bn[id] = b;
        parent[c.node] = id; parent[d.node] = id;
        uint32_t count, meta; float cost;
        merge_meta(prm, nmeta[c.node], nmeta[d.node], ncount[c.node], ncount[d.node], ncost[c.node], ncost[d.node], half_area(b.lo, b.hi), count, meta, cost);
        ncount[id] = count; nmeta[id] = meta; ncost[id] = cost;
        for (int k = 0; k < 3; k++) { c.lo[k] = b.lo[k]; c.hi[k] = b.hi[k]; }
        c.node = id;
    }
    cout[pos] = c;
}

// The TOP of the tree: PLOC stops at a few thousand clusters and the host's binned-SAH builder (zr_bvh.cpp) arranges them — a
// bottom-up merger is weakest where a top-down split is strongest, in the few levels every ray crosses.  The host sends only the
// topology (children of every new node, ids ascending by height so that a level is an id range = a batch); boxes, counts and the
// leaf decision come from the children here, exactly as at a PLOC merge.
__global__ __launch_bounds__(256) void k_top_nodes(const uint2* __restrict__ kids, uint32_t id_lo, uint32_t id_hi, uint32_t kid_base, Params prm, BNode* __restrict__ bn,
                                                   uint32_t* __restrict__ parent, uint32_t* __restrict__ ncount, uint32_t* __restrict__ nmeta, float* __restrict__ ncost) {
    const uint32_t id = id_lo + blockIdx.x * 256 + threadIdx.x;
    if (id >= id_hi) return;
    const uint2 k = kids[id - kid_base];
    const BNode a = bn[k.x], c = bn[k.y];
    BNode b;
    for (int q = 0; q < 3; q++) { b.lo[q] = fminf(a.lo[q], c.lo[q]); b.hi[q] = fmaxf(a.hi[q], c.hi[q]); }
    b.left = k.x; b.right = k.y;
    bn[id] = b;
    parent[k.x] = id; parent[k.y] = id;
    uint32_t count, meta; float cost;
    merge_meta(prm, nmeta[k.x], nmeta[k.y], ncount[k.x], ncount[k.y], ncost[k.x], ncost[k.y], half_area(b.lo, b.hi), count, meta, cost);
    ncount[id] = count; nmeta[id] = meta; ncost[id] = cost;
}

// ---- 4. order -----------------------------------------------------------------------------------------------------------
// A FINAL leaf is a leaf-flagged node whose parent is not (or the root).  It climbs to the root: depth-first position of its first
// primitive = the counts of every left sibling on the way; then lays its primitives out there.
__global__ __launch_bounds__(256) void k_order(const BNode* __restrict__ bn, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ ncount,
                                               const uint32_t* __restrict__ nmeta, uint32_t n_nodes, uint32_t* __restrict__ leaf_pos, uint32_t* __restrict__ dfs_obj,
                                               uint8_t* __restrict__ dfs_kind, uint32_t* __restrict__ result /* [0] max depth, [1] final leaves */) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
    uint32_t depth = 0;
    bool fin = false;
    if (id < n_nodes && (nmeta[id] & M_LEAF)) {
        const uint32_t p = parent[id];
        fin = p == NONE || !(nmeta[p] & M_LEAF);
    }
    if (fin) {
        uint32_t first = 0, cur = id;
        for (uint32_t p = parent[cur]; p != NONE; p = parent[cur]) {
            const BNode b = bn[p];
            if (b.right == cur) first += ncount[b.left];
            cur = p; depth++;
        }
        leaf_pos[id] = first;
        // the subtree's primitives, depth-first (left before right); at most max_leaf of them
        const uint32_t kind = nmeta[id] & M_KIND;
        uint32_t stack[24]; int sp = 0; uint32_t at = first;
        stack[sp++] = id;
        while (sp > 0) {
            const uint32_t q = stack[--sp];
            const BNode b = bn[q];
            if (b.right == NONE) { dfs_obj[at] = b.left; dfs_kind[at] = (uint8_t)kind; at++; }
            else if (sp <= 22) { stack[sp++] = b.right; stack[sp++] = b.left; }
        }
    }
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)depth, m, 64); depth = o > depth ? o : depth; }
    const unsigned long long bf = __ballot(fin);
    if ((threadIdx.x & 63) == 0 && bf != 0ull) { atomicMax(&result[0], depth); atomicAdd(&result[1], (uint32_t)__popcll(bf)); }
}

struct IsKind {
    uint8_t k;
    __host__ __device__ uint32_t operator()(uint8_t v) const { return v == k ? 1u : 0u; }
};
__global__ __launch_bounds__(256) void k_pick_rank(const uint8_t* __restrict__ dfs_kind, const uint32_t* __restrict__ scanned, uint32_t n, uint8_t k, uint32_t* __restrict__ rank) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p < n && dfs_kind[p] == k) rank[p] = scanned[p];
}
__global__ __launch_bounds__(256) void k_iota(uint32_t* __restrict__ a, uint32_t n) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p < n) a[p] = p;
}
// a final leaf's first primitive index in its kind's array: base[kind] + rank of its first primitive
__global__ __launch_bounds__(256) void k_leaf_first(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ nmeta, uint32_t n_nodes, const uint32_t* __restrict__ rank,
                                                    BuildPrimOut out, uint32_t* __restrict__ leaf_pos) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n_nodes || !(nmeta[id] & M_LEAF)) return;
    const uint32_t p = parent[id];
    if (p != NONE && (nmeta[p] & M_LEAF)) return;
    leaf_pos[id] = out.base[nmeta[id] & M_KIND] + rank[leaf_pos[id]];
}

// ---- 5. 4-wide nodes ------------------------------------------------------------------------------------------------------
// origin + q * scale against a plane x, as real numbers: a = origin and m = q * scale are exact doubles, TwoSum gives a + m = s + e
__device__ __forceinline__ void plane_sum(float origin, float scale, int q, double& s, double& e) {
    const double a = (double)origin, m = (double)q * (double)scale;
    s = a + m;
    const double bb = s - a;
    e = (a - (s - bb)) + (m - bb);
}
__device__ __forceinline__ bool plane_le(float origin, float scale, int q, double x) { double s, e; plane_sum(origin, scale, q, s, e); return s < x || (s == x && e <= 0.0); }
__device__ __forceinline__ bool plane_ge(float origin, float scale, int q, double x) { double s, e; plane_sum(origin, scale, q, s, e); return s > x || (s == x && e >= 0.0); }
__device__ __forceinline__ double plane_val(float origin, float scale, int q) { return (double)origin + (double)q * (double)scale; }

__device__ bool quant_axis(const float* lo, const float* hi, int n, float& origin, float& scale, uint32_t* qlo, uint32_t* qhi) {
    float mn = lo[0], mx = hi[0];
    for (int k = 1; k < n; k++) { mn = fminf(mn, lo[k]); mx = fmaxf(mx, hi[k]); }
    if (!(fabsf(mn) <= 1e30f) || !(fabsf(mx) <= 1e30f)) return false;
    origin = nextafterf(mn, -__builtin_huge_valf());
    const double ext = (double)mx - (double)origin;
    int e = ext > 0 ? (int)ceil(log2(ext / 255.0)) : -100;
    const int emin = origin != 0.0f ? max(-100, ilogbf(origin) - 30) : -100;
    if (e < emin) e = emin;
    for (int tries = 0; tries < 64; tries++, e++) {
        scale = ldexpf(1.0f, e);
        bool ok = true;
        for (int k = 0; k < n && ok; k++) {
            long ql = (long)floor(((double)lo[k] - (double)origin) / (double)scale);
            ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
            while (ql > 0 && !plane_le(origin, scale, (int)ql, (double)lo[k])) ql--;
            if (!plane_le(origin, scale, (int)ql, (double)lo[k])) ok = false;
            long qh = (long)ceil(((double)hi[k] - (double)origin) / (double)scale);
            qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
            while (qh < 255 && !plane_ge(origin, scale, (int)qh, (double)hi[k])) qh++;
            if (!plane_ge(origin, scale, (int)qh, (double)hi[k])) ok = false;
            qlo[k] = (uint32_t)ql; qhi[k] = (uint32_t)qh;
        }
        if (ok) return true;
    }
    return false;
}

// quantises the boxes of `kids` into nq; false when a box cannot be represented; worst = largest area inflation
__device__ bool quantise(const BNode* __restrict__ bn, const uint32_t* kids, int nk, NodeQ& nq, float& worst) {
    uint32_t ql[3][4], qh[3][4];
    for (int k = 0; k < 6; k++) nq.q[k] = 0;
    for (int ax = 0; ax < 3; ax++) {
        float lo[4], hi[4];
        for (int k = 0; k < nk; k++) { lo[k] = bn[kids[k]].lo[ax]; hi[k] = bn[kids[k]].hi[ax]; }
        if (!quant_axis(lo, hi, nk, nq.origin[ax], nq.scale[ax], ql[ax], qh[ax])) return false;
        for (int k = 0; k < nk; k++) { nq.q[ax] |= ql[ax][k] << (8 * k); nq.q[3 + ax] |= qh[ax][k] << (8 * k); }
    }
    worst = 1;
    for (int k = 0; k < nk; k++) {
        double d[3];
        for (int ax = 0; ax < 3; ax++) d[ax] = plane_val(nq.origin[ax], nq.scale[ax], (int)qh[ax][k]) - plane_val(nq.origin[ax], nq.scale[ax], (int)ql[ax][k]);
        const double aq = d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
        const BNode b = bn[kids[k]];
        const double at = (double)half_area(b.lo, b.hi);
        const float rr = at > 0 ? (float)(aq / at) : (aq > 0 ? 3.0e38f : 1.0f);
        if (!(rr <= worst)) worst = rr;
    }
    return true;
}

struct PlanOut {
    uint32_t counter;       // next free quad index
    uint32_t quant_fail;    // a box could not be put on a grid
    uint32_t quad_depth;    // deepest quad level
    uint32_t kept_closed;
    uint32_t root_demand;
    uint32_t pad_[3];
    NodeF root;
};

__device__ __forceinline__ uint32_t leaf_ref(const uint32_t* __restrict__ nmeta, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ leaf_pos, uint32_t node) {
    return ZR_REF_LEAF | ((nmeta[node] & M_KIND) << 28) | ((ncount[node] - 1u) << 24) | leaf_pos[node];
}

// One thread per node of a batch; nodes that are quad roots (qroot[id] assigned by their parent's plan, or the tree's root) plan
// their node.  `first_node == root` && !root_in_array: the root's boxes stay FP32 (PlanOut::root).
// qlevel[quad] = level of the quad (root's children = 1).  qinst[quad] = 1 + the largest stack demand of a placed group among the
// quad's instance leaves (0: none).
__global__ __launch_bounds__(256) void k_plan(const BNode* __restrict__ bn, const uint32_t* __restrict__ nmeta, const uint32_t* __restrict__ ncount,
                                              const uint32_t* __restrict__ leaf_pos, uint32_t node_lo, uint32_t node_hi, uint32_t root, int root_in_array, float open_ratio,
                                              const zr_object* __restrict__ objs, const uint32_t* __restrict__ run_demand,
                                              uint32_t* __restrict__ qroot, uint32_t* __restrict__ qlevel, uint32_t* __restrict__ qinst, NodeQ* __restrict__ quads,
                                              PlanOut* __restrict__ po) {
    const uint32_t id = node_lo + blockIdx.x * 256 + threadIdx.x;
    if (id >= node_hi) return;
    const bool is_root = id == root;
    const bool fp32_root = is_root && !root_in_array;
    const uint32_t me = qroot[id];
    if (!fp32_root && me == NONE) return;
    uint32_t kids[4]; int nk = 0;
    NodeQ nq;
    for (int ax = 0; ax < 3; ax++) { nq.origin[ax] = 0; nq.scale[ax] = 1; }
    for (int k = 0; k < 6; k++) nq.q[k] = 0;
    float worst = 1;
    if (nmeta[id] & M_LEAF) {   // the whole tree is one leaf (only the root can be a leaf AND a quad root)
        kids[nk++] = id;
        if (!fp32_root && !quantise(bn, kids, nk, nq, worst)) atomicExch(&po->quant_fail, 1u);
    } else {
        const BNode b = bn[id];
        kids[0] = b.left; kids[1] = b.right; nk = 2;
        if (!fp32_root && !quantise(bn, kids, nk, nq, worst)) atomicExch(&po->quant_fail, 1u);
        while (nk < 4) {
            // open the inner child with the largest area, unless the grid of the wider node would be too coarse for one of the boxes
            int best = -1; float ba = -1;
            for (int k = 0; k < nk; k++) {
                if (nmeta[kids[k]] & M_LEAF) continue;
                const BNode c = bn[kids[k]];
                const float ar = half_area(c.lo, c.hi);
                if (ar > ba) { ba = ar; best = k; }
            }
            if (best < 0) break;
            uint32_t trial[4];
            for (int k = 0; k < nk; k++) trial[k] = kids[k];
            const BNode ob = bn[kids[best]];
            trial[best] = ob.left; trial[nk] = ob.right;
            if (!fp32_root) {
                NodeQ tq = nq; float w = 1;
                if (!quantise(bn, trial, nk + 1, tq, w)) { atomicExch(&po->quant_fail, 1u); break; }
                if (w > open_ratio && w > worst) { atomicAdd(&po->kept_closed, 1u); break; }
                nq = tq; worst = w;
            }
            for (int k = 0; k <= nk; k++) kids[k] = trial[k];
            nk++;
        }
    }
    // references: leaves name their primitives, inner children get consecutive new quads
    int n_inner = 0;
    for (int k = 0; k < nk; k++) if (!(nmeta[kids[k]] & M_LEAF)) n_inner++;
    uint32_t next = n_inner ? atomicAdd(&po->counter, (uint32_t)n_inner) : 0u;
    const uint32_t my_level = fp32_root ? 0u : qlevel[me];
    uint32_t refs[4] = {ZR_REF_EMPTY, ZR_REF_EMPTY, ZR_REF_EMPTY, ZR_REF_EMPTY};
    uint32_t inst_demand = 0;
    for (int k = 0; k < nk; k++) {
        const uint32_t c = kids[k];
        if (nmeta[c] & M_LEAF) {
            refs[k] = leaf_ref(nmeta, ncount, leaf_pos, c);
            if ((nmeta[c] & M_KIND) == ZR_KIND_INSTANCE && run_demand && objs) {   // one placement per leaf: the primitive under c
                uint32_t q = c;
                while (bn[q].right != NONE) q = bn[q].left;
                const uint32_t g = objs[bn[q].left].index;
                const uint32_t d = 1u + run_demand[g];
                if (d > inst_demand) inst_demand = d;
            }
        } else { refs[k] = next; qroot[c] = next; qlevel[next] = my_level + 1u; next++; }
    }
    if (n_inner) atomicMax(&po->quad_depth, my_level + 1u);
    if (fp32_root) {
        NodeF rf;
        for (int k = 0; k < 4; k++) { rf.lox[k] = rf.loy[k] = rf.loz[k] = rf.hix[k] = rf.hiy[k] = rf.hiz[k] = 0.0f; rf.ref[k] = refs[k]; }
        for (int k = 0; k < nk; k++) {
            const BNode c = bn[kids[k]];
            rf.lox[k] = nextafterf(c.lo[0], -__builtin_huge_valf()); rf.loy[k] = nextafterf(c.lo[1], -__builtin_huge_valf()); rf.loz[k] = nextafterf(c.lo[2], -__builtin_huge_valf());
            rf.hix[k] = nextafterf(c.hi[0], __builtin_huge_valf()); rf.hiy[k] = nextafterf(c.hi[1], __builtin_huge_valf()); rf.hiz[k] = nextafterf(c.hi[2], __builtin_huge_valf());
        }
        po->root = rf;
        po->root_demand = inst_demand;   // (folded into the root's demand by k_demand_root)
    } else {
        for (int k = 0; k < 4; k++) nq.ref[k] = refs[k];
        quads[me] = nq;
        qinst[me] = inst_demand;
    }
}

// DETERMINISTIC QUAD NUMBERS.  k_plan hands out quad indices with an atomic counter: the frame does not depend on them, but the node array's layout — and
// with it the cache behaviour of EXTEND, a few per cent of noise between two commits of the same scene — did.  After the plan the quads are renumbered by the id of the
// binary node each one is rooted at, highest id first (ids are a pure function of the input; a parent's id is above its children's, so the order is top-down): an
// exclusive scan over "is a quad root" along descending ids gives the new number, the records move there with their inner references rewritten.
struct QRootFlag {
    const uint32_t* qroot; uint32_t top;
    __host__ __device__ uint32_t operator()(uint32_t k) const { return qroot[top - k] != 0xFFFFFFFFu ? 1u : 0u; }
};
__global__ __launch_bounds__(256) void k_qmap(uint32_t* __restrict__ qroot, const uint32_t* __restrict__ qscan, uint32_t n_nodes, uint32_t* __restrict__ qmap) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_nodes) return;
    const uint32_t id = n_nodes - 1 - k, old = qroot[id];
    if (old == NONE) return;
    qmap[old] = qscan[k];
    qroot[id] = qscan[k];
}
__global__ __launch_bounds__(256) void k_qpermute(const NodeQ* __restrict__ src, NodeQ* __restrict__ dst, const uint32_t* __restrict__ qinst_src, uint32_t* __restrict__ qinst_dst,
                                                  const uint32_t* __restrict__ qmap, uint32_t cap, PlanOut* __restrict__ po) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q == 0)   // the FP32 root's references (unused when the root is a stored node)
        for (int c = 0; c < 4; c++) { const uint32_t r = po->root.ref[c]; if (r != ZR_REF_EMPTY && !(r & ZR_REF_LEAF) && r < cap) po->root.ref[c] = qmap[r]; }
    if (q >= po->counter || q >= cap) return;
    NodeQ nq = src[q];
    for (int c = 0; c < 4; c++) { const uint32_t r = nq.ref[c]; if (r != ZR_REF_EMPTY && !(r & ZR_REF_LEAF)) nq.ref[c] = qmap[r]; }
    const uint32_t to = qmap[q];
    dst[to] = nq;
    qinst_dst[to] = qinst_src[q];
}

// forward over the batches (children before parents): demand(node) = max(nk, nk - 1 + max over inner children of their demand,
// nk - 1 + an instance leaf's 1 + its group's demand)  — Flattener::demand_of
__device__ __forceinline__ uint32_t demand_from(const uint32_t refs[4], uint32_t inst_demand, const uint32_t* __restrict__ qdem) {
    uint32_t nk = 0, best = inst_demand;
    for (int k = 0; k < 4; k++) {
        if (refs[k] == ZR_REF_EMPTY) continue;
        nk++;
        if (!(refs[k] & ZR_REF_LEAF)) { const uint32_t d = qdem[refs[k]]; if (d > best) best = d; }
    }
    const uint32_t deep = nk ? nk - 1u + best : 0u;
    return nk > deep ? nk : deep;
}
__global__ __launch_bounds__(256) void k_demand(const uint32_t* __restrict__ qroot, uint32_t node_lo, uint32_t node_hi, uint32_t root, int root_in_array,
                                                const NodeQ* __restrict__ quads, const uint32_t* __restrict__ qinst, uint32_t* __restrict__ qdem, PlanOut* __restrict__ po) {
    const uint32_t id = node_lo + blockIdx.x * 256 + threadIdx.x;
    if (id >= node_hi) return;
    if (id == root && !root_in_array) { po->root_demand = demand_from(po->root.ref, po->root_demand, qdem); return; }
    const uint32_t me = qroot[id];
    if (me == NONE) return;
    const uint32_t d = demand_from(quads[me].ref, qinst[me], qdem);
    qdem[me] = d;
    if (id == root) po->root_demand = d;
}

// ---- 6. sibling-pair records ------------------------------------------------------------------------------------------------
// inner (not leaf-flagged) nodes numbered from the root down the id range: rank of node id = exclusive count of inner nodes with
// a larger id (the root is pair 0)
struct InnerFlag {
    const uint32_t* nmeta; uint32_t top;   // top = id of the root = 2n - 2
    __host__ __device__ uint32_t operator()(uint32_t k) const { return (nmeta[top - k] & M_LEAF) ? 0u : 1u; }
};
__global__ __launch_bounds__(256) void k_pairs(const BNode* __restrict__ bn, const uint32_t* __restrict__ nmeta, const uint32_t* __restrict__ ncount,
                                               const uint32_t* __restrict__ leaf_pos, const uint32_t* __restrict__ prank, uint32_t n, NodePair* __restrict__ pairs) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;   // position from the top: node = (2n - 2) - k
    if (k + 1 >= n) return;
    const uint32_t top = 2 * n - 2, id = top - k;
    if (nmeta[id] & M_LEAF) return;
    const BNode b = bn[id];
    NodePair pr;
    const uint32_t ch[2] = {b.left, b.right};
    for (int s = 0; s < 2; s++) {
        const BNode c = bn[ch[s]];
        for (int a = 0; a < 3; a++) { pr.lo[s][a] = nextafterf(c.lo[a], -__builtin_huge_valf()); pr.hi[s][a] = nextafterf(c.hi[a], __builtin_huge_valf()); }
        if (nmeta[ch[s]] & M_LEAF) { pr.child[s] = leaf_pos[ch[s]]; pr.meta[s] = (((nmeta[ch[s]] & M_KIND) + 1u) << 16) | ncount[ch[s]]; }
        else { pr.child[s] = prank[top - ch[s]]; pr.meta[s] = 0; }
    }
    pairs[prank[k]] = pr;
}
// the whole tree in one leaf: a pair whose second child is empty (Flattener::fill_leaf_root)
__global__ void k_pair_leaf_root(const BNode* __restrict__ bn, const uint32_t* __restrict__ nmeta, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ leaf_pos,
                                 uint32_t root, NodePair* __restrict__ pairs) {
    const BNode c = bn[root];
    NodePair pr;
    for (int a = 0; a < 3; a++) { pr.lo[0][a] = nextafterf(c.lo[a], -__builtin_huge_valf()); pr.hi[0][a] = nextafterf(c.hi[a], __builtin_huge_valf()); pr.lo[1][a] = 0; pr.hi[1][a] = 0; }
    pr.child[0] = leaf_pos[root]; pr.meta[0] = (((nmeta[root] & M_KIND) + 1u) << 16) | ncount[root];
    pr.child[1] = 0; pr.meta[1] = (1u << 16) | 0u;
    pairs[0] = pr;
}

// ---- 7. primitive records (the host's Flattener::put_* rules) ------------------------------------------------------------------
__device__ void put_triangle_raw(const BuildPrimOut& out, size_t di, const double* v, const double* nn, uint32_t mat, bool force_front) {
    double* tv = out.tri_v + di * ZR_TRI_STRIDE;
    double* t = out.tri_s + di * 20;
    for (int k = 0; k < 9; k++) { tv[k] = v[k]; t[k] = v[k]; t[9 + k] = nn[k]; }
    t[18] = __longlong_as_double((long long)(unsigned long long)mat);
    t[19] = __longlong_as_double(force_front ? 1ll : 0ll);
}
__global__ __launch_bounds__(256) void k_emit(BuildSceneIn in, const zr_object* __restrict__ objs, const uint8_t* __restrict__ code, uint32_t first_triangle, uint32_t n,
                                              const uint32_t* __restrict__ dfs_obj, const uint8_t* __restrict__ dfs_kind, const uint32_t* __restrict__ rank, BuildPrimOut out,
                                              uint32_t* __restrict__ compound, uint32_t* __restrict__ n_compound) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t oi = dfs_obj[p], kind = dfs_kind[p];
    const size_t di = (size_t)out.base[kind] + rank[p];
    zr_object o; uint32_t bk = 0;
    if (objs) { o = objs[oi]; bk = code[oi] >> 4; }
    else { o.type = ZR_PRIM_TRIANGLE; o.index = first_triangle + oi; o.chain_first = 0; o.chain_count = 0; }
    if (kind == ZR_PRIM_MEDIUM || kind == ZR_KIND_WRAPPED) {   // compound: the host finishes these few (inner primitives behind the leaf ranges)
        const uint32_t at = atomicAdd(n_compound, 1u);
        compound[2 * at] = oi; compound[2 * at + 1] = (uint32_t)di;
        return;
    }
    if (o.type == ZR_PRIM_GROUP) {
        DInstance inst; inst.chain_first = o.chain_first; inst.chain_count = o.chain_count; inst.root = 0; inst.pad_ = 0;
        out.insts[di] = inst; out.inst_group[di] = o.index;
        return;
    }
    if (bk == 1) {   // put_baked_triangle: a triangle under translate / rotate_* / material_instance wrappers, stored in world space
        double v[9], nn[9];
        for (int k = 0; k < 9; k++) { v[k] = in.tri_v[(size_t)o.index * 9 + k]; nn[k] = in.tri_n[(size_t)o.index * 9 + k]; }
        uint32_t mat = in.tri_mat[o.index];
        bool force_front = false;
        for (int k = (int)o.chain_count - 1; k >= 0; k--) {
            const zr_xform_op op = in.ops[o.chain_first + k];
            const double sn = op.a[0], co = op.a[1];
            for (int c = 0; c < 3; c++) {
                double* pp = v + 3 * c; double* q = nn + 3 * c;
                switch (op.kind) {
                    case ZR_OP_TRANSLATE: pp[0] += op.a[0]; pp[1] += op.a[1]; pp[2] += op.a[2]; break;
                    case ZR_OP_ROTATE_Y: { double x = pp[0], z = pp[2]; pp[0] = co * x - sn * z; pp[2] = sn * x + co * z;
                                           x = q[0]; z = q[2]; q[0] = co * x - sn * z; q[2] = sn * x + co * z; } break;
                    case ZR_OP_ROTATE_X: { double y = pp[1], z = pp[2]; pp[1] = co * y - sn * z; pp[2] = sn * y + co * z;
                                           y = q[1]; z = q[2]; q[1] = co * y - sn * z; q[2] = sn * y + co * z; } break;
                    case ZR_OP_ROTATE_Z: { double x = pp[0], y = pp[1]; pp[0] = co * x - sn * y; pp[1] = sn * x + co * y;
                                           x = q[0]; y = q[1]; q[0] = co * x - sn * y; q[1] = sn * x + co * y; } break;
                    default: break;
                }
            }
            if (op.kind == ZR_OP_TRANSLATE || op.kind == ZR_OP_ROTATE_Y) force_front = true;
            if (op.kind == ZR_OP_MATERIAL) mat = op.mat;
        }
        put_triangle_raw(out, di, v, nn, mat, force_front);
    } else if (bk == 3) {   // put_baked_sphere: uniform scale / translate / material_instance
        const double* q = in.spheres + (size_t)o.index * 4;
        double c[3] = {q[0], q[1], q[2]}, r = fmax(0.0, q[3]);
        uint32_t mat = in.sphere_mat[o.index];
        bool force_front = false;
        for (int k = (int)o.chain_count - 1; k >= 0; k--) {
            const zr_xform_op op = in.ops[o.chain_first + k];
            if (op.kind == ZR_OP_SCALE) { c[0] *= op.a[0]; c[1] *= op.a[0]; c[2] *= op.a[0]; r *= op.a[0]; }
            else if (op.kind == ZR_OP_TRANSLATE) { c[0] += op.a[0]; c[1] += op.a[1]; c[2] += op.a[2]; force_front = true; }
            else if (op.kind == ZR_OP_MATERIAL) mat = op.mat;
        }
        double* d = out.spheres + di * 4;
        d[0] = c[0]; d[1] = c[1]; d[2] = c[2]; d[3] = r;
        out.sphere_mat[di] = force_front ? (mat | 0x80000000u) : mat;
    } else if (bk == 4) {   // put_pcube: cube -> [scale] -> [rotate_y] -> translate in one record
        const double* q = in.cubes + (size_t)o.index * 12;
        double rec[ZR_PCUBE_STRIDE] = {q[0], q[1], q[2], q[3], q[4], q[5], 0, 0, 0, 0, 1, 0, 1, 1, 1, 0};
        uint32_t mat = in.cube_mat[o.index];
        for (int k = (int)o.chain_count - 1; k >= 0; k--) {
            const zr_xform_op op = in.ops[o.chain_first + k];
            if (op.kind == ZR_OP_TRANSLATE) { rec[6] = op.a[0]; rec[7] = op.a[1]; rec[8] = op.a[2]; }
            else if (op.kind == ZR_OP_ROTATE_Y) { rec[9] = op.a[0]; rec[10] = op.a[1]; rec[11] = 1.0; }
            else if (op.kind == ZR_OP_SCALE) { rec[12] = op.a[0]; rec[13] = op.a[1]; rec[14] = op.a[2]; rec[15] = 1.0; }
            else if (op.kind == ZR_OP_MATERIAL) mat = op.mat;
        }
        for (int k = 0; k < ZR_PCUBE_STRIDE; k++) out.pcubes[di * ZR_PCUBE_STRIDE + k] = rec[k];
        out.pcube_mat[di] = mat;
    } else {
        const bool mat_only = bk == 2;   // material-only chain: the outermost wrapper is applied last
        if (o.type == ZR_PRIM_SPHERE) {
            const double* q = in.spheres + (size_t)o.index * 4;
            double* d = out.spheres + di * 4;
            d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = fmax(0.0, q[3]);   // sphere.hpp:9
            out.sphere_mat[di] = mat_only ? in.ops[o.chain_first].mat : in.sphere_mat[o.index];
        } else if (o.type == ZR_PRIM_TRIANGLE) {
            double v[9], nn[9];
            for (int k = 0; k < 9; k++) { v[k] = in.tri_v[(size_t)o.index * 9 + k]; nn[k] = in.tri_n[(size_t)o.index * 9 + k]; }
            put_triangle_raw(out, di, v, nn, in.tri_mat[o.index], false);
        } else {
            const double* q = in.cubes + (size_t)o.index * 12;
            for (int k = 0; k < 6; k++) out.cubes[di * 6 + k] = q[k];
            out.cube_mat[di] = mat_only ? in.ops[o.chain_first].mat : in.cube_mat[o.index];
        }
    }
}

// ---- relocation: a tree's local indices -> the scene's arrays ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reloc_pairs(const NodePair* __restrict__ src, uint32_t n, uint32_t off, NodePair* __restrict__ dst) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    NodePair p = src[i];
    for (int s = 0; s < 2; s++) if (p.meta[s] == 0) p.child[s] += off;
    dst[off + i] = p;
}
__global__ __launch_bounds__(256) void k_reloc_quads(const NodeQ* __restrict__ src, uint32_t n, uint32_t off, NodeQ* __restrict__ dst) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    NodeQ q = src[i];
    for (int k = 0; k < 4; k++) if (q.ref[k] != ZR_REF_EMPTY && !(q.ref[k] & ZR_REF_LEAF)) q.ref[k] += off;
    dst[off + i] = q;
}
__global__ __launch_bounds__(256) void k_patch_insts(DInstance* __restrict__ insts, const uint32_t* __restrict__ inst_group, uint32_t n, const uint32_t* __restrict__ run_root,
                                                     const uint32_t* __restrict__ run_qroot) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = inst_group[i];
    insts[i].root = run_root[g]; insts[i].pad_ = run_qroot[g];
}
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* __restrict__ a, size_t n, uint32_t v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = v;
}

inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + 255) / 256)); }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

DeviceBuilder::~DeviceBuilder() {
    if (arena_) (void)hipFree(arena_);
    for (void* p : keep_) (void)hipFree(p);
}
void* DeviceBuilder::alloc_keep(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes < 64 ? 64 : bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    keep_.push_back(p);
    return p;
}
size_t DeviceBuilder::scratch_bytes(uint32_t n) {
    const size_t N = 2 * (size_t)n + 2;
    // generous: the arrays listed in build() + rocPRIM's temporaries
    return (size_t)n * (32 + 16 + 8 + 64 + 4 + 1 + 4 + 1 + 4 + 4 + 4 + 4 + 4 + 4 + 8) + N * (32 + 4 + 4 + 4 + 4 + 4 + 4) + (size_t)n * 24 + (64u << 20);
}
hipError_t DeviceBuilder::reserve(size_t bytes) {
    if (bytes <= arena_bytes_) return hipSuccess;
    if (arena_) { (void)hipStreamSynchronize(st_); (void)hipFree(arena_); arena_ = nullptr; arena_bytes_ = 0; }
    hipError_t e = hipMalloc((void**)&arena_, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); err_ = "out of device memory for the build's scratch"; return e; }
    arena_bytes_ = bytes;
    return hipSuccess;
}

hipError_t DeviceBuilder::build(const BuildSceneIn& in, const zr_object* d_objects, const uint8_t* d_code, uint32_t first_triangle, uint32_t n,
                                const BuildParams& bp, bool root_in_array, const BuildPrimOut& out, const uint32_t* d_run_demand, uint32_t depth_limit,
                                bool stats, BuiltTree& t) {
    hipError_t e;
    err_ = ""; use_host_ = false;
    if (n == 0) { err_ = "empty tree"; use_host_ = true; return hipErrorInvalidValue; }
    Params prm;
    prm.ct = bp.ct; prm.max_leaf = bp.max_leaf < 1 ? 1 : (bp.max_leaf > 16 ? 16 : bp.max_leaf); prm.open_ratio = bp.open_ratio;
    prm.radius = bp.radius < 1 ? 1 : (bp.radius > MAX_R ? MAX_R : bp.radius);
    for (int k = 0; k < 8; k++) { prm.ck[k] = bp.ck[k]; prm.cap[k] = bp.leaf_cap[k]; }
    // ---- carve the arena (laid out twice: once from a null base to learn the size, once for real)
    const size_t N = 2 * (size_t)n;
    const uint32_t nb_max = (n + 255) / 256;
    PBox* pbox; uint64_t *keys_a, *keys_b; uint32_t *vals_a, *vals_b; Cl *cl_a, *cl_b; BNode* bn;
    uint32_t *parent, *ncount, *nmeta, *leaf_pos, *qroot, *nn, *blk_keep, *blk_new, *keep_off, *new_off, *dfs_obj, *rank, *scanned, *qlevel, *qinst, *qdem, *compound, *state;
    float* ncost; uint8_t *act, *dfs_kind; PlanOut* po; void* tmp; size_t tmp_bytes = 0;
    uint32_t *qinst_plan, *qmap, *qscan; NodeQ* quads_plan;
    auto layout = [&](unsigned char* base) -> size_t {
        unsigned char* at = base;
        auto take = [&](size_t bytes) { void* p = at; at += (bytes + 255) / 256 * 256; return p; };
        pbox = (PBox*)take((size_t)n * sizeof(PBox));
        keys_a = (uint64_t*)take((size_t)n * 8); keys_b = (uint64_t*)take((size_t)n * 8);
        vals_a = (uint32_t*)take((size_t)n * 4); vals_b = (uint32_t*)take((size_t)n * 4);
        cl_a = (Cl*)take((size_t)n * sizeof(Cl)); cl_b = (Cl*)take((size_t)n * sizeof(Cl));
        bn = (BNode*)take(N * sizeof(BNode));
        parent = (uint32_t*)take(N * 4); ncount = (uint32_t*)take(N * 4); nmeta = (uint32_t*)take(N * 4);
        ncost = (float*)take(N * 4);
        leaf_pos = (uint32_t*)take(N * 4); qroot = (uint32_t*)take(N * 4);
        nn = (uint32_t*)take((size_t)n * 4); act = (uint8_t*)take(n);
        blk_keep = (uint32_t*)take((size_t)nb_max * 4); blk_new = (uint32_t*)take((size_t)nb_max * 4);
        keep_off = (uint32_t*)take((size_t)nb_max * 4); new_off = (uint32_t*)take((size_t)nb_max * 4);
        dfs_obj = (uint32_t*)take((size_t)n * 4); dfs_kind = (uint8_t*)take(n);
        rank = (uint32_t*)take((size_t)n * 4); scanned = (uint32_t*)take((size_t)n * 4);
        qlevel = (uint32_t*)take((size_t)n * 4); qinst = (uint32_t*)take((size_t)n * 4); qdem = (uint32_t*)take((size_t)n * 4);
        qinst_plan = (uint32_t*)take((size_t)n * 4); qmap = (uint32_t*)take((size_t)n * 4); qscan = (uint32_t*)take(N * 4);
        quads_plan = (NodeQ*)take((size_t)n * sizeof(NodeQ));   // the plan's records under the atomic counter's numbers, before k_qpermute
        compound = (uint32_t*)take((size_t)n * 8);
        state = (uint32_t*)take(256);        // [0] survivors, [1] new nodes; [8..14] bounds; [16] max depth, [17] final leaves; [20] compound count
        po = (PlanOut*)take(sizeof(PlanOut));
        if (tmp_bytes == 0) {   // rocPRIM temporaries: the largest of the sort's and the two scans'
            size_t a = 0, b = 0, c = 0;
            (void)rocprim::radix_sort_pairs(nullptr, a, keys_a, keys_b, vals_a, vals_b, (size_t)n, 0, 63, st_);
            (void)rocprim::exclusive_scan(nullptr, b, rocprim::make_transform_iterator(dfs_kind, IsKind{0}), scanned, 0u, (size_t)n, rocprim::plus<uint32_t>(), st_);
            (void)rocprim::exclusive_scan(nullptr, c, rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), InnerFlag{nmeta, 0}), scanned, 0u,
                                          (size_t)n, rocprim::plus<uint32_t>(), st_);
            size_t d = 0;
            (void)rocprim::exclusive_scan(nullptr, d, rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), QRootFlag{qroot, 0}), qscan, 0u, N,
                                          rocprim::plus<uint32_t>(), st_);
            if (d > c) c = d;
            tmp_bytes = (a > b ? (a > c ? a : c) : (b > c ? b : c)) + 4096;
        }
        tmp = take(tmp_bytes);
        return (size_t)(at - base);
    };
    const size_t need = layout(nullptr);
    if ((e = reserve(need)) != hipSuccess) return e;
    (void)layout(arena_);
    double t0 = now_s();
    auto lap = [&](int k) { if (stats) { (void)hipStreamSynchronize(st_); const double t1 = now_s(); t.ms[k] += (t1 - t0) * 1e3; t0 = t1; } };

    // ---- 1 + 2: boxes, keys, sort
    uint32_t h_bounds[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0};
    if ((e = hipMemcpyAsync(state + 8, h_bounds, sizeof h_bounds, hipMemcpyHostToDevice, st_)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_boxes, grid_for(n), dim3(256), 0, st_, in, d_objects, d_code, first_triangle, n, pbox, state + 8);
    hipLaunchKernelGGL(k_keys, grid_for(n), dim3(256), 0, st_, pbox, n, state + 8, keys_a, vals_a);
    lap(0);
    if (t.want_boxes) {
        t.dbg_boxes.resize((size_t)n * 8);
        if ((e = hipMemcpyAsync(t.dbg_boxes.data(), pbox, (size_t)n * sizeof(PBox), hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
    }
    if ((e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_a, keys_b, vals_a, vals_b, (size_t)n, 0, 63, st_)) != hipSuccess) { err_ = "radix sort failed"; return e; }
    lap(1);
    // ---- 3: PLOC
    if ((e = hipMemsetAsync(parent, 0xFF, N * 4, st_)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_init_clusters, grid_for(n), dim3(256), 0, st_, pbox, vals_b, n, prm, cl_a, bn, ncount, nmeta, ncost);
    std::vector<uint32_t> batch_start;   // inner-node ids: batch b = [batch_start[b], batch_start[b + 1])
    batch_start.push_back(n);
    uint32_t n_cur = n, next_node = n;
    Cl* cin = cl_a; Cl* cout = cl_b;
    uint32_t h_state[8];
    if ((e = hipMemcpyAsync(h_state, state + 8, 8 * 4, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
    if (h_state[6] != 0) { err_ = "an object's box is not finite or beyond 1e18: host builder"; use_host_ = true; return hipErrorInvalidValue; }
    const uint32_t top = (bp.top_clusters > 1 && (uint64_t)n >= 8ull * (uint64_t)bp.top_clusters) ? (uint32_t)bp.top_clusters : 1u;
    while (n_cur > top) {
        const uint32_t nb = (n_cur + 255) / 256;
        hipLaunchKernelGGL(k_ploc_nn, dim3(nb), dim3(256), 0, st_, cin, n_cur, prm.radius, nn, act, blk_keep, blk_new);
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st_, blk_keep, blk_new, nb, keep_off, new_off, state);
        hipLaunchKernelGGL(k_ploc_write, dim3(nb), dim3(256), 0, st_, cin, n_cur, nn, act, keep_off, new_off, next_node, prm, cout, bn, parent, ncount, nmeta, ncost);
        if ((e = hipMemcpyAsync(h_state, state, 8, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
        const uint32_t survivors = h_state[0], created = h_state[1];
        if (created == 0 || survivors + created != n_cur) { err_ = "PLOC made no progress (internal error)"; return hipErrorUnknown; }
        next_node += created; n_cur = survivors;
        batch_start.push_back(next_node);
        Cl* sw = cin; cin = cout; cout = sw;
        t.ploc_iterations++;
    }
    if (n_cur > 1) {   // the top of the tree over the remaining clusters, by the host's SAH builder
        std::vector<Cl> hc(n_cur);
        if ((e = hipMemcpyAsync(hc.data(), cin, (size_t)n_cur * sizeof(Cl), hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
        std::vector<BuildBox> tb(n_cur);
        std::vector<uint32_t> tk(n_cur, 0u);
        for (uint32_t k = 0; k < n_cur; k++) for (int a = 0; a < 3; a++) { tb[k].lo[a] = hc[k].lo[a]; tb[k].hi[a] = hc[k].hi[a]; }
        BuildResult tr;
        const double ck1[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        build_bvh(tb, tk, 1, 26, 1.0, ck1, tr);   // one cluster per leaf; the depth budget leaves the clusters' own subtrees theirs
        // heights (a leaf = a cluster = height 0), then the inner nodes in ascending height: new id = next_node + position
        struct TopNode { uint32_t id; int height; };
        std::vector<TopNode> inner;
        std::vector<int> height(tr.nodes.size(), -1);
        {
            std::vector<std::pair<uint32_t, bool>> stk;   // post-order walk without recursion
            stk.emplace_back(0u, false);
            while (!stk.empty()) {
                auto [id, done] = stk.back(); stk.pop_back();
                const BuildNode& nd = tr.nodes[id];
                if (nd.count) { height[id] = 0; continue; }
                if (!done) { stk.emplace_back(id, true); stk.emplace_back((uint32_t)nd.left, false); stk.emplace_back((uint32_t)nd.right, false); }
                else { height[id] = 1 + std::max(height[nd.left], height[nd.right]); inner.push_back({id, height[id]}); }
            }
        }
        std::stable_sort(inner.begin(), inner.end(), [](const TopNode& a, const TopNode& b) { return a.height < b.height; });
        std::vector<uint32_t> new_id(tr.nodes.size(), NONE);
        for (size_t k = 0; k < inner.size(); k++) new_id[inner[k].id] = next_node + (uint32_t)k;
        auto node_of = [&](int32_t id) { const BuildNode& nd = tr.nodes[id]; return nd.count ? hc[tr.order[nd.first]].node : new_id[id]; };
        std::vector<uint2> kids(inner.size());
        for (size_t k = 0; k < inner.size(); k++) { const BuildNode& nd = tr.nodes[inner[k].id]; kids[k].x = node_of(nd.left); kids[k].y = node_of(nd.right); }
        if (inner.size() + 1 != (size_t)n_cur) { err_ = "top tree: node count mismatch (internal error)"; return hipErrorUnknown; }
        uint2* d_kids = (uint2*)nn;   // (free now: n_cur - 1 pairs of ids fit the n words of the neighbour array)
        if ((e = hipMemcpyAsync(d_kids, kids.data(), kids.size() * sizeof(uint2), hipMemcpyHostToDevice, st_)) != hipSuccess) return e;
        const uint32_t kid_base = next_node;
        for (size_t k = 0; k < inner.size();) {   // one launch per height level: its children are complete
            size_t k1 = k;
            while (k1 < inner.size() && inner[k1].height == inner[k].height) k1++;
            const uint32_t lo = next_node + (uint32_t)k, hi = next_node + (uint32_t)k1;
            hipLaunchKernelGGL(k_top_nodes, grid_for(hi - lo), dim3(256), 0, st_, d_kids, lo, hi, kid_base, prm, bn, parent, ncount, nmeta, ncost);
            batch_start.push_back(hi);
            k = k1;
        }
        if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;   // (kids[] is read by the copy until here)
        next_node += (uint32_t)inner.size();
        n_cur = 1;
    }
    lap(2);
    const uint32_t n_nodes = next_node;                  // = 2n - 1
    const uint32_t root = n_nodes - 1;
    // ---- 4: order
    if ((e = hipMemsetAsync(state + 16, 0, 8 * 4, st_)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_order, grid_for(n_nodes), dim3(256), 0, st_, bn, parent, ncount, nmeta, n_nodes, leaf_pos, dfs_obj, dfs_kind, state + 16);
    // per-kind ranks of the depth-first sequence; the kinds present come back with the counts
    uint32_t h_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        // count per kind: one scan per kind over the sequence, the last element's inclusive value is the kind's total
        // (kinds are few: a world of bare triangles runs one scan, or none — a single kind's rank is the position itself)
        std::vector<uint8_t> present;
        // which kinds exist: the caller's codes told the host already, but a group's tree has no codes: find out from the root's meta
        uint32_t h_root_meta = 0;
        if ((e = hipMemcpyAsync(&h_root_meta, nmeta + root, 4, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(h_state, state + 16, 8, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
        t.depth = h_state[0];
        if (t.depth >= depth_limit) { err_ = "device tree deeper than the traversal stack allows: host builder"; use_host_ = true; return hipErrorInvalidValue; }
        if (h_root_meta & M_UNIFORM) {
            const uint32_t k = h_root_meta & M_KIND;
            h_cnt[k] = n;
            hipLaunchKernelGGL(k_iota, grid_for(n), dim3(256), 0, st_, rank, n);
        } else {
            for (uint8_t k = 0; k < 7; k++) {
                auto flags = rocprim::make_transform_iterator(dfs_kind, IsKind{k});
                if ((e = rocprim::exclusive_scan(tmp, tmp_bytes, flags, scanned, 0u, (size_t)n, rocprim::plus<uint32_t>(), st_)) != hipSuccess) { err_ = "scan failed"; return e; }
                uint32_t last_rank = 0; uint8_t last_kind = 0;
                if ((e = hipMemcpyAsync(&last_rank, scanned + (n - 1), 4, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
                if ((e = hipMemcpyAsync(&last_kind, dfs_kind + (n - 1), 1, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
                if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
                h_cnt[k] = last_rank + (last_kind == k ? 1u : 0u);
                if (h_cnt[k]) hipLaunchKernelGGL(k_pick_rank, grid_for(n), dim3(256), 0, st_, dfs_kind, scanned, n, k, rank);
            }
        }
    }
    for (int k = 0; k < 8; k++) t.cnt[k] = h_cnt[k];
    hipLaunchKernelGGL(k_leaf_first, grid_for(n_nodes), dim3(256), 0, st_, parent, nmeta, n_nodes, rank, out, leaf_pos);
    lap(3);
    // ---- 5: 4-wide nodes — top-down = the batches in reverse
    const uint32_t quad_cap = n;   // at most one quad per inner node
    NodeQ* quads = (NodeQ*)alloc_keep((size_t)quad_cap * sizeof(NodeQ));
    NodePair* pairs = (NodePair*)alloc_keep((size_t)n * sizeof(NodePair));
    if (!quads || !pairs) { err_ = "out of device memory for the tree's records"; return hipErrorOutOfMemory; }
    hipLaunchKernelGGL(k_fill_u32, grid_for(N), dim3(256), 0, st_, qroot, N, NONE);
    PlanOut h_po_in; std::memset(&h_po_in, 0, sizeof h_po_in);   // (source of an asynchronous copy: not written again before the next synchronisation)
    h_po_in.counter = root_in_array ? 1u : 0u;
    if ((e = hipMemcpyAsync(po, &h_po_in, sizeof h_po_in, hipMemcpyHostToDevice, st_)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(qinst_plan, 0, (size_t)n * 4, st_)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(qinst, 0, (size_t)n * 4, st_)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(qdem, 0, (size_t)n * 4, st_)) != hipSuccess) return e;
    auto renumber = [&]() -> hipError_t {   // see k_qmap
        auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), QRootFlag{qroot, root});
        hipError_t er = rocprim::exclusive_scan(tmp, tmp_bytes, flags, qscan, 0u, (size_t)n_nodes, rocprim::plus<uint32_t>(), st_);
        if (er != hipSuccess) { err_ = "scan failed"; return er; }
        hipLaunchKernelGGL(k_qmap, grid_for(n_nodes), dim3(256), 0, st_, qroot, qscan, n_nodes, qmap);
        hipLaunchKernelGGL(k_qpermute, grid_for(quad_cap), dim3(256), 0, st_, quads_plan, quads, qinst_plan, qinst, qmap, quad_cap, po);
        return hipSuccess;
    };
    if (root_in_array) {   // the root is quad 0, level 0
        if ((e = hipMemsetAsync(qroot + root, 0, 4, st_)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(qlevel, 0, 4, st_)) != hipSuccess) return e;
    }
    if (n == 1) {   // a single primitive: the "batch" is the primitive itself
        hipLaunchKernelGGL(k_plan, dim3(1), dim3(256), 0, st_, bn, nmeta, ncount, leaf_pos, root, root + 1, root, root_in_array ? 1 : 0, prm.open_ratio, d_objects, d_run_demand,
                           qroot, qlevel, qinst_plan, quads_plan, po);
        if ((e = renumber()) != hipSuccess) return e;
        hipLaunchKernelGGL(k_demand, dim3(1), dim3(256), 0, st_, qroot, root, root + 1, root, root_in_array ? 1 : 0, quads, qinst, qdem, po);
    } else {
        for (size_t b = batch_start.size() - 1; b-- > 0;) {
            const uint32_t lo = batch_start[b], hi = batch_start[b + 1];
            hipLaunchKernelGGL(k_plan, grid_for(hi - lo), dim3(256), 0, st_, bn, nmeta, ncount, leaf_pos, lo, hi, root, root_in_array ? 1 : 0, prm.open_ratio, d_objects,
                               d_run_demand, qroot, qlevel, qinst_plan, quads_plan, po);
        }
        if ((e = renumber()) != hipSuccess) return e;
        for (size_t b = 0; b + 1 < batch_start.size(); b++) {
            const uint32_t lo = batch_start[b], hi = batch_start[b + 1];
            hipLaunchKernelGGL(k_demand, grid_for(hi - lo), dim3(256), 0, st_, qroot, lo, hi, root, root_in_array ? 1 : 0, quads, qinst, qdem, po);
        }
    }
    lap(4);
    // ---- 6: pair records
    uint32_t n_pairs = 1;
    if (n >= 2) {
        uint32_t h_root_meta = 0;
        if ((e = hipMemcpyAsync(&h_root_meta, nmeta + root, 4, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
        if (h_root_meta & M_LEAF) hipLaunchKernelGGL(k_pair_leaf_root, dim3(1), dim3(1), 0, st_, bn, nmeta, ncount, leaf_pos, root, pairs);
        else {
            auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), InnerFlag{nmeta, root});
            uint32_t* prank = scanned;   // (free again)
            if ((e = rocprim::exclusive_scan(tmp, tmp_bytes, flags, prank, 0u, (size_t)(n - 1), rocprim::plus<uint32_t>(), st_)) != hipSuccess) { err_ = "scan failed"; return e; }
            hipLaunchKernelGGL(k_pairs, grid_for(n - 1), dim3(256), 0, st_, bn, nmeta, ncount, leaf_pos, prank, n, pairs);
            uint32_t last = 0, last_meta = 0;   // total = rank of the lowest inner id + its own flag
            if ((e = hipMemcpyAsync(&last, prank + (n - 2), 4, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
            if ((e = hipMemcpyAsync(&last_meta, nmeta + n, 4, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
            n_pairs = last + ((last_meta & M_LEAF) ? 0u : 1u);
        }
    } else hipLaunchKernelGGL(k_pair_leaf_root, dim3(1), dim3(1), 0, st_, bn, nmeta, ncount, leaf_pos, root, pairs);
    lap(5);
    // ---- 7: primitive records
    if ((e = hipMemsetAsync(state + 20, 0, 4, st_)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_emit, grid_for(n), dim3(256), 0, st_, in, d_objects, d_code, first_triangle, n, dfs_obj, dfs_kind, rank, out, compound, state + 20);
    uint32_t n_compound = 0;
    BNode h_root;
    PlanOut h_po;
    if ((e = hipMemcpyAsync(&h_po, po, sizeof h_po, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(&n_compound, state + 20, 4, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(&h_root, bn + root, sizeof h_root, hipMemcpyDeviceToHost, st_)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st_)) != hipSuccess) return e;
    t.compound.resize((size_t)n_compound * 2);
    if (n_compound && (e = hipMemcpy(t.compound.data(), compound, (size_t)n_compound * 8, hipMemcpyDeviceToHost)) != hipSuccess) return e;
    lap(6);
    t.pairs = pairs; t.n_pairs = n_pairs; t.quads = quads; t.n_quads = h_po.counter;
    t.root = h_po.root; t.quad_depth = h_po.quad_depth; t.demand = h_po.root_demand; t.quant_ok = h_po.quant_fail == 0;
    for (int k = 0; k < 3; k++) { t.box[k] = h_root.lo[k]; t.box[3 + k] = h_root.hi[k]; }
    if ((e = hipGetLastError()) != hipSuccess) { err_ = "a build kernel failed to launch"; return e; }
    return hipSuccess;
}

hipError_t DeviceBuilder::relocate(const BuiltTree& t, NodePair* pairs_dst, uint32_t pair_off, NodeQ* quads_dst, uint32_t quad_off) {
    if (t.n_pairs) hipLaunchKernelGGL(k_reloc_pairs, grid_for(t.n_pairs), dim3(256), 0, st_, t.pairs, t.n_pairs, pair_off, pairs_dst);
    if (t.n_quads) hipLaunchKernelGGL(k_reloc_quads, grid_for(t.n_quads), dim3(256), 0, st_, t.quads, t.n_quads, quad_off, quads_dst);
    return hipGetLastError();
}
hipError_t DeviceBuilder::patch_instances(DInstance* insts, const uint32_t* inst_group, uint32_t n, const uint32_t* d_run_root, const uint32_t* d_run_qroot) {
    if (n) hipLaunchKernelGGL(k_patch_insts, grid_for(n), dim3(256), 0, st_, insts, inst_group, n, d_run_root, d_run_qroot);
    return hipGetLastError();
}

}  // namespace zr
