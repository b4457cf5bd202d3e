// zr_bvh.cpp — binned-SAH BVH2 builder, multi-threaded over the top of the tree.  See zr_bvh.h.
#include "zr_bvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <limits>
#include <thread>

namespace zr {
namespace {

constexpr int kBins = 16;
const double kInf = std::numeric_limits<double>::infinity();

inline void grow(BuildBox& a, const BuildBox& b) {
    for (int k = 0; k < 3; k++) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); }
}
inline BuildBox empty_box() { return BuildBox{{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}}; }
inline double half_area(const BuildBox& b) {
    double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0;
    // clamp astronomically large extents (ground spheres of radius 1000 are fine, infinities are not)
    return dx * dy + dy * dz + dz * dx;
}
inline int ceil_log2(uint32_t n) { int l = 0; while ((1u << l) < n) l++; return l; }

struct Builder {
    const std::vector<BuildBox>& boxes;
    const std::vector<uint32_t>& kinds;
    int max_leaf, depth_limit;
    int leaf_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per kind; 0 = max_leaf
    int cap_of(uint32_t kind) const { const int c = leaf_cap[kind & 7]; return c > 0 && c < max_leaf ? c : max_leaf; }
    double ct;
    const double* ck;
    std::vector<BuildNode> nodes;
    std::atomic<uint32_t> n_nodes{0};
    std::vector<uint32_t> order;
    std::atomic<int> max_depth{0};
    std::atomic<int> threads_left{0};

    Builder(const std::vector<BuildBox>& b, const std::vector<uint32_t>& k) : boxes(b), kinds(k) {}

    uint32_t alloc() { return n_nodes.fetch_add(1); }

    bool homogeneous(uint32_t first, uint32_t count) const {
        uint32_t k0 = kinds[order[first]];
        for (uint32_t i = 1; i < count; i++) if (kinds[order[first + i]] != k0) return false;
        return true;
    }

    void note_depth(int d) { int cur = max_depth.load(); while (d > cur && !max_depth.compare_exchange_weak(cur, d)) {} }

    void build(uint32_t id, uint32_t first, uint32_t count, int depth) {
        BuildNode& n = nodes[id];
        n.box = empty_box();
        BuildBox cb = empty_box();
        for (uint32_t i = 0; i < count; i++) {
            const BuildBox& b = boxes[order[first + i]];
            grow(n.box, b);
            for (int k = 0; k < 3; k++) { double c = 0.5 * (b.lo[k] + b.hi[k]); cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c); }
        }
        const bool homo = homogeneous(first, count);
        auto make_leaf = [&]() { n.first = first; n.count = count; n.kind = kinds[order[first]]; n.left = n.right = -1; note_depth(depth); };
        if (count == 1) { make_leaf(); return; }

        // forced balanced splits when the depth budget is nearly used up
        const bool forced = depth + ceil_log2(count) + 2 >= depth_limit;
        uint32_t mid = 0;
        int axis = 0;
        {
            double ext[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
            if (ext[1] > ext[axis]) axis = 1;
            if (ext[2] > ext[axis]) axis = 2;
        }
        bool have_split = false;
        if (!forced) {
            // binned SAH over the three axes
            double best = kInf; int best_axis = -1, best_bin = -1;
            const double leaf_cost_each = ck[kinds[order[first]] & 7];
            for (int ax = 0; ax < 3; ax++) {
                double lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0) || !std::isfinite(ext)) continue;
                BuildBox bb[kBins]; uint32_t bc[kBins]; double bw[kBins];
                for (int b = 0; b < kBins; b++) { bb[b] = empty_box(); bc[b] = 0; bw[b] = 0; }
                double scale = kBins / ext;
                for (uint32_t i = 0; i < count; i++) {
                    uint32_t o = order[first + i];
                    const BuildBox& b = boxes[o];
                    int bi = (int)((0.5 * (b.lo[ax] + b.hi[ax]) - lo) * scale);
                    bi = bi < 0 ? 0 : (bi >= kBins ? kBins - 1 : bi);
                    grow(bb[bi], b); bc[bi]++; bw[bi] += ck[kinds[o] & 7];
                }
                double la[kBins], lw[kBins]; BuildBox acc = empty_box(); double w = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    if (bc[b]) grow(acc, bb[b]);
                    w += bw[b]; la[b] = half_area(acc); lw[b] = w;
                }
                acc = empty_box(); w = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    if (bc[b]) grow(acc, bb[b]);
                    w += bw[b];
                    if (lw[b - 1] == 0 || w == 0) continue;
                    double cost = la[b - 1] * lw[b - 1] + half_area(acc) * w;
                    if (cost < best) { best = cost; best_axis = ax; best_bin = b; }
                }
            }
            double pa = half_area(n.box);
            if (best_axis >= 0) {
                double split_cost = ct * pa + best;
                double leaf_cost = 0;
                for (uint32_t i = 0; i < count; i++) leaf_cost += ck[kinds[order[first + i]] & 7];
                leaf_cost *= pa; (void)leaf_cost_each;
                if (homo && (int)count <= cap_of(kinds[order[first]]) && leaf_cost <= split_cost) { make_leaf(); return; }
                double lo = cb.lo[best_axis], scale = kBins / (cb.hi[best_axis] - cb.lo[best_axis]);
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t o) {
                    const BuildBox& b = boxes[o];
                    int bi = (int)((0.5 * (b.lo[best_axis] + b.hi[best_axis]) - lo) * scale);
                    bi = bi < 0 ? 0 : (bi >= kBins ? kBins - 1 : bi);
                    return bi < best_bin;
                });
                mid = (uint32_t)(it - order.begin());
                have_split = mid > first && mid < first + count;
            } else if (homo && (int)count <= cap_of(kinds[order[first]])) {
                make_leaf(); return;  // all centroids coincide
            }
        } else if (homo && (int)count <= cap_of(kinds[order[first]])) {
            make_leaf(); return;
        }
        if (!have_split) {
            // object median along the widest centroid axis; if centroids coincide, split mixed kinds apart
            mid = first + count / 2;
            if (cb.hi[axis] - cb.lo[axis] > 0) {
                std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t a, uint32_t b) {
                    return boxes[a].lo[axis] + boxes[a].hi[axis] < boxes[b].lo[axis] + boxes[b].hi[axis];
                });
            } else if (!homo) {
                std::sort(order.begin() + first, order.begin() + first + count, [&](uint32_t a, uint32_t b) { return kinds[a] < kinds[b]; });
                uint32_t k0 = kinds[order[first]];
                mid = first; while (kinds[order[mid]] == k0) mid++;
            }
        }
        uint32_t l = alloc(), r = alloc();
        nodes[id].left = (int32_t)l; nodes[id].right = (int32_t)r; nodes[id].count = 0;
        uint32_t lc = mid - first, rc = first + count - mid;
        if (lc > 32768 && rc > 32768 && threads_left.fetch_sub(1) > 0) {
            std::thread t([&, l, first, lc, depth]() { build(l, first, lc, depth + 1); });
            build(r, mid, rc, depth + 1);
            t.join();
            threads_left.fetch_add(1);
        } else {
            build(l, first, lc, depth + 1);
            build(r, mid, rc, depth + 1);
        }
    }
};

}  // namespace

void build_bvh(const std::vector<BuildBox>& boxes, const std::vector<uint32_t>& kinds, int max_leaf, int depth_limit,
               double cost_traverse, const double cost_kind[8], BuildResult& out, const int* max_leaf_kind) {
    out.nodes.clear(); out.order.clear(); out.max_depth = 0;
    const uint32_t n = (uint32_t)boxes.size();
    if (n == 0) return;
    Builder b(boxes, kinds);
    b.max_leaf = std::max(1, std::min(max_leaf, 0xFFFF));
    b.depth_limit = depth_limit; b.ct = cost_traverse; b.ck = cost_kind;
    if (max_leaf_kind) for (int k = 0; k < 8; k++) b.leaf_cap[k] = max_leaf_kind[k];
    b.nodes.resize((size_t)2 * n);
    b.order.resize(n);
    for (uint32_t i = 0; i < n; i++) b.order[i] = i;
    unsigned hw = std::thread::hardware_concurrency();
    b.threads_left = (int)std::min(16u, hw > 1 ? hw - 1 : 0u);
    uint32_t root = b.alloc();
    b.build(root, 0, n, 0);
    b.nodes.resize(b.n_nodes.load());
    out.nodes.swap(b.nodes);
    out.order.swap(b.order);
    out.max_depth = b.max_depth.load();
}

}  // namespace zr
