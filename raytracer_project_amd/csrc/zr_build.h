// zr_build.h — host-visible interface of the DEVICE-side BVH build (zr_build.hip).  Replaces bvh_node's constructor
// (/root/reference/bvh.hpp:11-44), which the reference runs on every render restart (main.cpp:1492-1500), with a build that
// never leaves the GPU: object boxes -> Morton keys -> radix sort -> PLOC merging (parallel locally-ordered clustering: every
// cluster looks for its nearest neighbour, by merged surface area, within a window of the Morton order; mutual pairs merge;
// repeat) -> SAH leaf collapse -> the 2->4 collapse onto 8-bit quantised 64-byte nodes -> sibling-pair records -> the primitive
// records in leaf order.  The output is what the host path (zr_bvh.cpp + Flattener in zr_flatten.h) produces: the same arrays,
// another (equally valid) tree.  Closest hit does not depend on the tree.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <vector>

#include "zr_device_types.h"

namespace zr {

// the scene as the caller gave it, uploaded unchanged
struct BuildSceneIn {
    const double* spheres = nullptr; const uint32_t* sphere_mat = nullptr;
    const double* tri_v = nullptr; const double* tri_n = nullptr; const uint32_t* tri_mat = nullptr;
    const double* cubes = nullptr; const uint32_t* cube_mat = nullptr;
    const zr_medium* media = nullptr; const zr_xform_op* ops = nullptr;
    const double* group_box = nullptr;   // per zr_group: lo[3], hi[3] of its triangles in their own space
};

struct BuildParams {
    float ct = 1.0f;           // cost of a node visit
    float ck[8] = {1, 1.5f, 1, 3, 3, 1.5f, 16, 1};   // cost of testing one primitive of each leaf kind
    int max_leaf = 4;
    int leaf_cap[8] = {0, 0, 1, 1, 1, 1, 1, 0};      // per kind; 0 = max_leaf
    float open_ratio = 1.25f;  // 4-wide collapse: a child is not opened when the wider node's grid would inflate a box's area beyond this
    int radius = 16;           // PLOC search radius (clusters on either side in Morton order)
    int top_clusters = 16384;  // PLOC stops at this many clusters and the host's binned-SAH builder arranges them (0: PLOC to the root); worlds under 8 x this: PLOC alone.
                               // zr_commit.cpp picks n / 64 within 4096 ... 65536
};

// where a tree's leaves put their primitive records: the scene's final arrays and the first index this tree may use per leaf kind
struct BuildPrimOut {
    double* spheres = nullptr; uint32_t* sphere_mat = nullptr;
    double* tri_v = nullptr; double* tri_s = nullptr;
    double* cubes = nullptr; uint32_t* cube_mat = nullptr;
    double* pcubes = nullptr; uint32_t* pcube_mat = nullptr;
    DInstance* insts = nullptr; uint32_t* inst_group = nullptr;
    uint32_t base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// a built tree: device records with indices local to the tree (relocate() moves them into the scene's arrays)
struct BuiltTree {
    NodePair* pairs = nullptr; uint32_t n_pairs = 0;
    NodeQ* quads = nullptr; uint32_t n_quads = 0;
    NodeF root{};                   // world tree: the root of the 4-wide tree (kernel arguments); unused when the root is a stored node
    double box[6] = {0, 0, 0, 0, 0, 0};   // the root's box (from outward-rounded floats)
    uint32_t depth = 0;             // binary depth of the deepest leaf
    uint32_t quad_depth = 0;
    uint32_t demand = 0;            // worst-case entries on an EXTEND lane's stack (Flattener::demand_of)
    bool quant_ok = true;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // leaf primitives per kind
    std::vector<uint32_t> compound; // (object, index in its kind's array) pairs of the MEDIUM / WRAPPED leaves: finished by the host
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // phase times when stats are on: boxes+keys, sort, merge, order, plan, pairs, emit, (spare)
    uint32_t ploc_iterations = 0;
    bool want_boxes = false;        // self-check (ZR_BUILD_CHECK): the objects' boxes as the device computed them, 8 floats each (lo, kind, hi, pad)
    std::vector<float> dbg_boxes;
};

// Scratch arena + the build itself.  One instance per commit; all work is queued on `stream`, the calls synchronise it where the
// host needs a count back.
class DeviceBuilder {
public:
    explicit DeviceBuilder(hipStream_t stream) : st_(stream) {}
    ~DeviceBuilder();
    DeviceBuilder(const DeviceBuilder&) = delete;
    DeviceBuilder& operator=(const DeviceBuilder&) = delete;
    // bytes of scratch a tree over n objects needs (the arena grows on demand; reserving up front saves reallocations)
    static size_t scratch_bytes(uint32_t n);
    hipError_t reserve(size_t bytes);
    // Builds the tree over world-list entries d_objects[0 .. n) (device), d_code[k] = leaf kind | baked << 4 of entry k.
    // d_objects == nullptr: the entries are the bare triangles first_triangle + k (a zr_group's run).
    // root_in_array: the root becomes quads[0] (a group's tree) instead of BuiltTree::root.
    // d_run_demand: per group its tree's stack demand (world tree with placements), may be null.
    // An input outside what this builder handles — non-finite or > 1e18 coordinates, a tree deeper than depth_limit (n < 2 is handled) — makes build()
    // return an error AND wants_host() true: the caller then uses the host builder.  Any other error is a genuine failure (error() says which), except that a
    // caller may also treat hipErrorOutOfMemory as "use the host builder": the device build keeps the scene as given, its arena and the final arrays alive at
    // once, several times the host path's footprint.
    hipError_t build(const BuildSceneIn& in, const zr_object* d_objects, const uint8_t* d_code, uint32_t first_triangle, uint32_t n,
                     const BuildParams& prm, bool root_in_array, const BuildPrimOut& out, const uint32_t* d_run_demand,
                     uint32_t depth_limit, bool stats, BuiltTree& t);
    // copies a built tree's pair records / 4-wide nodes into the scene's arrays at pair_off / quad_off, moving inner references
    hipError_t relocate(const BuiltTree& t, NodePair* pairs_dst, uint32_t pair_off, NodeQ* quads_dst, uint32_t quad_off);
    // placements: root / qroot of every instance from its group (after the groups' offsets are known)
    hipError_t patch_instances(DInstance* insts, const uint32_t* inst_group, uint32_t n, const uint32_t* d_run_root, const uint32_t* d_run_qroot);
    const char* error() const { return err_; }
    bool wants_host() const { return use_host_; }
    // memory the built trees live in until relocate(): owned by the builder
    void* alloc_keep(size_t bytes);
private:
    hipStream_t st_;
    unsigned char* arena_ = nullptr; size_t arena_bytes_ = 0;
    std::vector<void*> keep_;
    const char* err_ = "";
    bool use_host_ = false;
};

}  // namespace zr
