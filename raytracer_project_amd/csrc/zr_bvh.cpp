// zr_bvh.cpp — binned-SAH BVH2 builder for the host side of zr_scene_commit.  See zr_bvh.h.
//
// Built for commit latency (the reference rebuilds its BVH on every render restart, main.cpp:1492-1500): 1 M triangles in
// ~0.1 s on a few cores instead of the reference's 5-26 s (bvh.hpp:11-44: random axis, std::sort per node).
//   * object references are 48-byte records {float box, float centroid, id, kind} that are PHYSICALLY partitioned, so every
//     pass streams memory instead of gathering 48-byte double boxes through an index array;
//   * one pass bins all three axes (16 bins each); the split is the best of the 45 candidate planes;
//   * the top of the tree is built by teams of threads: a team splits its node together (chunked binning with per-thread bins,
//     a stable two-pass parallel partition), then divides itself between the two sides in proportion to their sizes and the
//     subtrees proceed concurrently; a team of one — or a node below kTeamNode references — builds serially;
//   * node boxes are exact: floats only rank split candidates; every node's box is the union of its objects' double boxes,
//     taken on the way back up.
// The tree is a pure function of the input (the parallel partition is stable, subtrees are independent), whatever the
// number of threads.
#include "zr_bvh.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

#include <emmintrin.h>   // SSE2: part of the x86-64 baseline

namespace zr {
namespace {

constexpr int kBins = 16;
constexpr uint32_t kTeamNode = 1u << 13;   // nodes with fewer references are not worth a team: the team's threads cost more than they save
constexpr uint32_t kSweep = 16;            // nodes up to this size: exact sweep over the sorted references instead of bins
const double kInf = std::numeric_limits<double>::infinity();
const float kInfF = std::numeric_limits<float>::infinity();

struct alignas(16) Ref {
    float lo[3]; uint32_t id;     // the object's box, rounded to nearest: ranks split candidates only
    float hi[3]; uint32_t kind;
};
static_assert(sizeof(Ref) == 32, "Ref is 32 bytes: two 16-byte loads");

// 4-float boxes on SSE2 (part of x86-64): lane 3 rides along unused
struct FBox {
    __m128 lo, hi;
    void clear() { lo = _mm_set1_ps(kInfF); hi = _mm_set1_ps(-kInfF); }
    void grow(__m128 l, __m128 h) { lo = _mm_min_ps(lo, l); hi = _mm_max_ps(hi, h); }
    void grow(const FBox& b) { grow(b.lo, b.hi); }
    void grow_point(__m128 p) { lo = _mm_min_ps(lo, p); hi = _mm_max_ps(hi, p); }
    float lo_of(int k) const { alignas(16) float v[4]; _mm_store_ps(v, lo); return v[k]; }
    float hi_of(int k) const { alignas(16) float v[4]; _mm_store_ps(v, hi); return v[k]; }
    float half_area() const {   // dx dy + dy dz + dz dx; 0 for an empty box.  Float: it only ranks candidates
        const __m128 d = _mm_sub_ps(hi, lo);                                    // dx dy dz ?
        const __m128 r = _mm_shuffle_ps(d, d, _MM_SHUFFLE(3, 0, 2, 1));         // dy dz dx ?
        const __m128 p = _mm_mul_ps(d, r);                                      // dxdy dydz dzdx ?
        const float a = _mm_cvtss_f32(_mm_add_ss(_mm_add_ss(p, _mm_shuffle_ps(p, p, 1)), _mm_shuffle_ps(p, p, 2)));
        return (_mm_movemask_ps(_mm_cmplt_ps(d, _mm_setzero_ps())) & 7) ? 0.0f : a;
    }
};
// lane 3 of the two loads holds id / kind: as floats those bit patterns are denormals, and arithmetic on denormals takes a
// microcode assist (~150 cycles) — the loads clear the lane
inline __m128 xyz_mask() { return _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1)); }
inline __m128 ref_lo(const Ref& r) { return _mm_and_ps(_mm_load_ps(r.lo), xyz_mask()); }
inline __m128 ref_hi(const Ref& r) { return _mm_and_ps(_mm_load_ps(r.hi), xyz_mask()); }
inline __m128 ref_centroid(const Ref& r) { return _mm_mul_ps(_mm_add_ps(ref_lo(r), ref_hi(r)), _mm_set1_ps(0.5f)); }
inline float centroid_of(const Ref& r, int a) { return 0.5f * (r.lo[a] + r.hi[a]); }

struct Bins {
    FBox box[3][kBins];
    uint32_t cnt[3][kBins];
    float w[3][kBins];   // sum of the kinds' test costs (small multiples of 0.5: exact in float far beyond any bin's population)
    void clear(int nb) { for (int a = 0; a < 3; a++) for (int b = 0; b < nb; b++) { box[a][b].clear(); cnt[a][b] = 0; w[a][b] = 0; } }
    void merge(const Bins& o, int nb) {
        for (int a = 0; a < 3; a++) for (int b = 0; b < nb; b++) { if (o.cnt[a][b]) box[a][b].grow(o.box[a][b]); cnt[a][b] += o.cnt[a][b]; w[a][b] += o.w[a][b]; }
    }
};

inline void grow(BuildBox& a, const BuildBox& b) {
    for (int k = 0; k < 3; k++) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); }
}
inline BuildBox empty_box() { return BuildBox{{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}}; }
inline int ceil_log2(uint32_t n) { int l = 0; while ((1u << l) < n) l++; return l; }

// runs fn(t) for t in [0, n) on n threads (the calling thread takes t = 0)
template <class F>
void team_run(int n, F&& fn) {
    if (n <= 1) { fn(0); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)n - 1);
    for (int t = 1; t < n; t++) th.emplace_back([&fn, t]() { fn(t); });
    fn(0);
    for (auto& t : th) t.join();
}

struct Builder {
    const std::vector<BuildBox>& boxes;
    int max_leaf = 4, depth_limit = 46;
    int leaf_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per kind; 0 = max_leaf
    double ct = 1.0;
    const double* ck = nullptr;
    float ckf[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    int team = 1;
    RawArray<Ref> refs, tmp;
    NodeArray nodes;
    std::atomic<int> max_depth{0};

    explicit Builder(const std::vector<BuildBox>& b) : boxes(b) {}

    int cap_of(uint32_t kind) const { const int c = leaf_cap[kind & 7]; return c > 0 && c < max_leaf ? c : max_leaf; }
    // node ids without a shared counter: the subtree over `count` references rooted at node `id` owns the ids
    // [id, id + 2 count - 1) — the left child is id + 1, the right child follows the left subtree's range.  A leaf of several
    // objects leaves the rest of its range unused (the array is not zero-filled: untouched pages cost nothing).
    static uint32_t left_id(uint32_t id) { return id + 1; }
    static uint32_t right_id(uint32_t id, uint32_t left_count) { return id + 2 * left_count; }
    void note_depth(int d) { int cur = max_depth.load(); while (d > cur && !max_depth.compare_exchange_weak(cur, d)) {} }

    bool homogeneous(uint32_t first, uint32_t count) const {
        const uint32_t k0 = refs[first].kind;
        for (uint32_t i = 1; i < count; i++) if (refs[first + i].kind != k0) return false;
        return true;
    }
    void make_leaf(uint32_t id, uint32_t first, uint32_t count, int depth) {
        BuildNode& n = nodes[id];
        n.first = first; n.count = count; n.kind = refs[first].kind; n.left = n.right = -1;
        n.box = empty_box();
        for (uint32_t i = 0; i < count; i++) grow(n.box, boxes[refs[first + i].id]);   // exact
        note_depth(depth);
    }

    static inline int bin_of(float c, float lo, float scale, int nb) {
        int b = (int)((c - lo) * scale);
        return b < 0 ? 0 : (b >= nb ? nb - 1 : b);
    }

    // centroid bounds of a range
    static void centroid_bounds(const Ref* r, uint32_t count, FBox& cb) {
        cb.clear();
        for (uint32_t i = 0; i < count; i++) cb.grow_point(ref_centroid(r[i]));
    }
    // one pass bins all three axes: bin index of every axis from one vector operation
    void bin_range(const Ref* r, uint32_t count, const FBox& cb, const float scale[3], int nb, Bins& bins) const {
        const __m128 cblo = cb.lo, sc = _mm_set_ps(0.0f, scale[2], scale[1], scale[0]);
        const __m128 top = _mm_set1_ps((float)nb - 0.5f), zero = _mm_setzero_ps();
        const bool use[3] = {scale[0] > 0, scale[1] > 0, scale[2] > 0};
        for (uint32_t i = 0; i < count; i++) {
            const Ref& q = r[i];
            const __m128 lo = ref_lo(q), hi = ref_hi(q);
            const __m128 c = _mm_mul_ps(_mm_add_ps(lo, hi), _mm_set1_ps(0.5f));
            const __m128i bi = _mm_cvttps_epi32(_mm_min_ps(_mm_max_ps(_mm_mul_ps(_mm_sub_ps(c, cblo), sc), zero), top));
            alignas(16) int b[4]; _mm_store_si128((__m128i*)b, bi);
            const float w = ckf[q.kind & 7];
            if (use[0]) { bins.box[0][b[0]].grow(lo, hi); bins.cnt[0][b[0]]++; bins.w[0][b[0]] += w; }
            if (use[1]) { bins.box[1][b[1]].grow(lo, hi); bins.cnt[1][b[1]]++; bins.w[1][b[1]] += w; }
            if (use[2]) { bins.box[2][b[2]].grow(lo, hi); bins.cnt[2][b[2]]++; bins.w[2][b[2]] += w; }
        }
    }

    struct Split { int axis = -1, bin = -1; double cost = kInf; double total_w = 0; FBox bounds; };
    static Split best_split(const Bins& bins, const float scale[3], int nb) {
        Split s;
        s.bounds.clear();
        bool have_total = false;
        for (int a = 0; a < 3; a++) {
            if (!(scale[a] > 0)) continue;
            double la[kBins], lw[kBins];
            FBox acc; acc.clear(); double w = 0;
            for (int b = 0; b < nb - 1; b++) {
                if (bins.cnt[a][b]) acc.grow(bins.box[a][b]);
                w += bins.w[a][b]; la[b] = acc.half_area(); lw[b] = w;
            }
            if (!have_total) {   // every axis bins every reference: totals from the first usable axis
                FBox all = acc; double tw = w;
                if (bins.cnt[a][nb - 1]) all.grow(bins.box[a][nb - 1]);
                tw += bins.w[a][nb - 1];
                s.bounds = all; s.total_w = tw; have_total = true;
            }
            acc.clear(); w = 0;
            for (int b = nb - 1; b > 0; b--) {
                if (bins.cnt[a][b]) acc.grow(bins.box[a][b]);
                w += bins.w[a][b];
                if (lw[b - 1] == 0 || w == 0) continue;
                const double cost = la[b - 1] * lw[b - 1] + (double)acc.half_area() * w;
                if (cost < s.cost) { s.cost = cost; s.axis = a; s.bin = b; }
            }
        }
        return s;
    }

    // ---- one node: choose the split and partition refs[first, first + count); returns mid (== first: became a leaf) ----
    // cb: bounds of the references' centroids (from the parent's partition pass); cbl / cbr: the same for the two sides.
    // `par` > 1: the whole team works on this node
    uint32_t split_node(uint32_t id, uint32_t first, uint32_t count, int depth, int par, const FBox& cb, FBox& cbl, FBox& cbr) {
        Ref* r = refs.data() + first;
        if (count == 1) { make_leaf(id, first, count, depth); return first; }
        const bool small = (int)count <= max_leaf;
        const bool homo = small && homogeneous(first, count);
        const bool may_leaf = homo && (int)count <= cap_of(r[0].kind);
        const bool forced = depth + ceil_log2(count) + 2 >= depth_limit;   // balanced splits when the depth budget is nearly used up
        float ext[3], scale[3];
        int widest = 0;
        // 16 bins per axis, fewer in small nodes (a bin per ~4 references): the fixed cost of a node — clearing the bins and
        // pricing 3 (nb - 1) planes — is what the bottom half of the tree is made of
        const int nb = count >= 64 ? kBins : std::max(4, (int)(count / 4));
        for (int a = 0; a < 3; a++) {
            ext[a] = cb.hi_of(a) - cb.lo_of(a);
            scale[a] = (ext[a] > 0 && std::isfinite(ext[a])) ? (float)nb / ext[a] : 0.0f;
            if (ext[a] > ext[widest]) widest = a;
        }
        uint32_t mid = first;
        bool have_split = false, have_cb = false;
        if (!forced && count <= kSweep) {
            // small node (most nodes of the tree): full sweep.  Per axis the references are sorted by centroid, every object
            // partition "first i | rest" is priced, and the node is reordered along the winning axis.
            uint8_t ord[3][kSweep];
            float cen[kSweep];
            double best = kInf; int bax = -1; uint32_t bi = 0; double total_w = 0; FBox all; all.clear();
            for (uint32_t i = 0; i < count; i++) { all.grow(ref_lo(r[i]), ref_hi(r[i])); total_w += ckf[r[i].kind & 7]; }
            for (int a = 0; a < 3; a++) {
                if (!(ext[a] > 0)) continue;
                uint8_t* o = ord[a];
                for (uint32_t i = 0; i < count; i++) cen[i] = centroid_of(r[i], a);
                for (uint32_t i = 0; i < count; i++) {   // insertion sort by (centroid, id): deterministic
                    uint32_t j = i;
                    while (j > 0 && (cen[o[j - 1]] > cen[i] || (cen[o[j - 1]] == cen[i] && r[o[j - 1]].id > r[i].id))) { o[j] = o[j - 1]; j--; }
                    o[j] = (uint8_t)i;
                }
                float ra[kSweep], rw[kSweep];
                FBox acc; acc.clear(); float w = 0;
                for (uint32_t i = count; i-- > 1;) { acc.grow(ref_lo(r[o[i]]), ref_hi(r[o[i]])); w += ckf[r[o[i]].kind & 7]; ra[i] = acc.half_area(); rw[i] = w; }
                acc.clear(); w = 0;
                for (uint32_t i = 1; i < count; i++) {
                    acc.grow(ref_lo(r[o[i - 1]]), ref_hi(r[o[i - 1]])); w += ckf[r[o[i - 1]].kind & 7];
                    const double cost = (double)acc.half_area() * w + (double)ra[i] * rw[i];
                    if (cost < best) { best = cost; bax = a; bi = i; }
                }
            }
            if (bax >= 0) {
                const double pa = all.half_area();
                if (may_leaf && total_w * pa <= ct * pa + best) { make_leaf(id, first, count, depth); return first; }
                Ref t[kSweep];
                for (uint32_t i = 0; i < count; i++) t[i] = r[ord[bax][i]];
                std::memcpy(r, t, (size_t)count * sizeof(Ref));
                mid = first + bi; have_split = true;
            } else if (may_leaf) { make_leaf(id, first, count, depth); return first; }
        } else if (!forced) {
            Bins bins; bins.clear(nb);
            if (par > 1) {
                std::vector<Bins> part((size_t)par);
                team_run(par, [&](int t) { part[t].clear(nb); const uint32_t a = (uint32_t)((uint64_t)count * t / par), b = (uint32_t)((uint64_t)count * (t + 1) / par); bin_range(r + a, b - a, cb, scale, nb, part[t]); });
                for (auto& p : part) bins.merge(p, nb);
            } else bin_range(r, count, cb, scale, nb, bins);
            const Split s = best_split(bins, scale, nb);
            if (s.axis >= 0) {
                const double pa = s.bounds.half_area();
                if (may_leaf && s.total_w * pa <= ct * pa + s.cost) { make_leaf(id, first, count, depth); return first; }
                const int ax = s.axis; const float lo = cb.lo_of(ax), sc = scale[ax]; const int sb = s.bin;
                auto left_of = [&](const Ref& q) { return bin_of(centroid_of(q, ax), lo, sc, nb) < sb; };
                if (par > 1) {   // stable parallel partition through tmp: counts, prefix, scatter (+ the sides' centroid bounds), copy back
                    std::vector<uint32_t> nl((size_t)par + 1, 0);
                    std::vector<FBox> pl((size_t)par), pr((size_t)par);
                    team_run(par, [&](int t) { const uint32_t a = (uint32_t)((uint64_t)count * t / par), b = (uint32_t)((uint64_t)count * (t + 1) / par); uint32_t c = 0; for (uint32_t i = a; i < b; i++) c += left_of(r[i]); nl[t + 1] = c; });
                    for (int t = 0; t < par; t++) nl[t + 1] += nl[t];
                    const uint32_t total_left = nl[par];
                    Ref* o = tmp.data() + first;
                    team_run(par, [&](int t) {
                        const uint32_t a = (uint32_t)((uint64_t)count * t / par), b = (uint32_t)((uint64_t)count * (t + 1) / par);
                        uint32_t li = nl[t], ri = total_left + (a - nl[t]);
                        FBox l, rr; l.clear(); rr.clear();
                        for (uint32_t i = a; i < b; i++) { if (left_of(r[i])) { l.grow_point(ref_centroid(r[i])); o[li++] = r[i]; } else { rr.grow_point(ref_centroid(r[i])); o[ri++] = r[i]; } }
                        pl[t] = l; pr[t] = rr;
                    });
                    team_run(par, [&](int t) { const uint32_t a = (uint32_t)((uint64_t)count * t / par), b = (uint32_t)((uint64_t)count * (t + 1) / par); std::memcpy(r + a, o + a, (size_t)(b - a) * sizeof(Ref)); });
                    cbl.clear(); cbr.clear();
                    for (int t = 0; t < par; t++) { cbl.grow(pl[t]); cbr.grow(pr[t]); }
                    mid = first + total_left;
                } else {   // in place, two cursors; the sides' centroid bounds come with the same pass
                    cbl.clear(); cbr.clear();
                    uint32_t i = 0, j = count;
                    for (;;) {
                        while (i < j && left_of(r[i])) { cbl.grow_point(ref_centroid(r[i])); i++; }
                        while (i < j && !left_of(r[j - 1])) { cbr.grow_point(ref_centroid(r[j - 1])); j--; }
                        if (i >= j) break;
                        std::swap(r[i], r[j - 1]);
                    }
                    mid = first + i;
                }
                have_split = mid > first && mid < first + count;
                have_cb = have_split;
            } else if (may_leaf) { make_leaf(id, first, count, depth); return first; }   // all centroids coincide
        } else if (may_leaf) { make_leaf(id, first, count, depth); return first; }
        if (!have_split) {
            // object median along the widest centroid axis; if the centroids coincide, split mixed kinds apart
            mid = first + count / 2;
            if (ext[widest] > 0) {
                std::nth_element(r, r + count / 2, r + count, [&](const Ref& a, const Ref& b) { const float ca = centroid_of(a, widest), cb2 = centroid_of(b, widest); return ca < cb2 || (ca == cb2 && a.id < b.id); });
            } else if (!(small ? homo : homogeneous(first, count))) {
                std::stable_sort(r, r + count, [](const Ref& a, const Ref& b) { return a.kind < b.kind; });
                const uint32_t k0 = r[0].kind;
                mid = first; while (refs[mid].kind == k0) mid++;
            }
        }
        if (!have_cb) { centroid_bounds(refs.data() + first, mid - first, cbl); centroid_bounds(refs.data() + mid, first + count - mid, cbr); }
        return mid;
    }

    // serial recursion below the team level
    void build(uint32_t id, uint32_t first, uint32_t count, int depth, const FBox& cb) {
        FBox cbl, cbr;
        const uint32_t mid = split_node(id, first, count, depth, 1, cb, cbl, cbr);
        if (mid == first) return;
        const uint32_t l = left_id(id), r = right_id(id, mid - first);
        nodes[id].left = (int32_t)l; nodes[id].right = (int32_t)r; nodes[id].count = 0; nodes[id].first = 0; nodes[id].kind = 0;
        build(l, first, mid - first, depth + 1, cbl);
        build(r, mid, first + count - mid, depth + 1, cbr);
        nodes[id].box = nodes[l].box; grow(nodes[id].box, nodes[r].box);
    }
    // top of the tree: a team of threads splits the node (chunked binning, stable parallel partition), then the team itself
    // splits — in proportion to the two sides — and the two subtrees proceed concurrently; a team of one builds serially.
    void build_team(uint32_t id, uint32_t first, uint32_t count, int depth, const FBox& cb, int tm) {
        if (tm <= 1 || count <= kTeamNode) { build(id, first, count, depth, cb); return; }
        FBox cbl, cbr;
        // splitting together pays from ~16K references per thread; below that the node is split by one thread and only the two
        // subtrees run concurrently
        const int par = count >= 4 * kTeamNode ? std::min(tm, (int)(count / (2 * kTeamNode))) : 1;
        const uint32_t mid = split_node(id, first, count, depth, par, cb, cbl, cbr);
        if (mid == first) return;
        const uint32_t l = left_id(id), r = right_id(id, mid - first);
        nodes[id].left = (int32_t)l; nodes[id].right = (int32_t)r; nodes[id].count = 0; nodes[id].first = 0; nodes[id].kind = 0;
        const uint32_t lc = mid - first, rc = first + count - mid;
        int tl = (int)((uint64_t)tm * lc / count);
        tl = tl < 1 ? 1 : (tl > tm - 1 ? tm - 1 : tl);
        std::thread other([&, this]() { build_team(l, first, lc, depth + 1, cbl, tl); });
        build_team(r, mid, rc, depth + 1, cbr, tm - tl);
        other.join();
        nodes[id].box = nodes[l].box; grow(nodes[id].box, nodes[r].box);
    }
};

}  // namespace

void build_bvh(const std::vector<BuildBox>& boxes, const std::vector<uint32_t>& kinds, int max_leaf, int depth_limit,
               double cost_traverse, const double cost_kind[8], BuildResult& out, const int* max_leaf_kind) {
    out.nodes.clear(); out.order.clear(); out.max_depth = 0;
    const uint32_t n = (uint32_t)boxes.size();
    if (n == 0) return;
    Builder b(boxes);
    b.max_leaf = std::max(1, std::min(max_leaf, 0xFFFF));
    b.depth_limit = depth_limit; b.ct = cost_traverse; b.ck = cost_kind;
    for (int k = 0; k < 8; k++) b.ckf[k] = (float)cost_kind[k];
    if (max_leaf_kind) for (int k = 0; k < 8; k++) b.leaf_cap[k] = max_leaf_kind[k];
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("ZR_BVH_THREADS")) hw = (unsigned)std::max(1, std::atoi(e));
    b.team = (int)std::max(1u, std::min(32u, hw));   // measured on the 256-thread host of an MI355X box, 1 M triangles: 16 threads 47 ms, 32: 40 ms, 64 and 128: 60 ms
    if (n < 64 * kTeamNode) b.team = std::max(1, std::min(b.team, (int)(n / kTeamNode)));
    // nothing below is value-initialised: pages are first touched by the threads that fill them
    b.nodes.allocate((size_t)2 * n);
    b.refs.allocate(n); b.tmp.allocate(b.team > 1 ? n : 0);
    std::vector<FBox> part((size_t)b.team);
    team_run(b.team, [&](int t) {
        const uint32_t a = (uint32_t)((uint64_t)n * t / b.team), e = (uint32_t)((uint64_t)n * (t + 1) / b.team);
        FBox cb; cb.clear();
        for (uint32_t i = a; i < e; i++) {
            Ref& r = b.refs[i];
            const BuildBox& bb = boxes[i];
            for (int k = 0; k < 3; k++) { r.lo[k] = (float)bb.lo[k]; r.hi[k] = (float)bb.hi[k]; }
            r.id = i; r.kind = kinds[i];
            cb.grow_point(ref_centroid(r));
        }
        part[t] = cb;
    });
    FBox cb; cb.clear();
    for (const FBox& p : part) cb.grow(p);
    static const bool prof = std::getenv("ZR_BVH_PROFILE") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    const uint32_t root = 0;
    b.nodes[root].left = b.nodes[root].right = -1; b.nodes[root].count = 0; b.nodes[root].first = 0; b.nodes[root].kind = 0;
    b.build_team(root, 0, n, 0, cb, b.team);
    double t1 = now(), t2 = t1;
    out.order.resize(n);
    team_run(b.team, [&](int t) {
        const uint32_t a = (uint32_t)((uint64_t)n * t / b.team), e = (uint32_t)((uint64_t)n * (t + 1) / b.team);
        for (uint32_t i = a; i < e; i++) out.order[i] = b.refs[i].id;
    });
    out.nodes = std::move(b.nodes);   // 2 n entries: ids are positions in the implicit layout, unused ones were never touched
    out.max_depth = b.max_depth.load();
    if (prof) std::fprintf(stderr, "[zr] bvh: tree %.1f ms, order %.1f ms, team %d\n", (t1 - t0) * 1e3, (now() - t2) * 1e3, b.team);
}

}  // namespace zr
