// zr_bvh.h — host-side BVH2 builder (binned SAH).  Replaces bvh_node's constructor
// (/root/reference/bvh.hpp:11-44: random axis, std::sort, median split, 26 s for 1M triangles) with a
// builder whose output is the flat sibling-pair array the kernels walk.  The tree may differ freely from
// the reference's: closest hit does not depend on it.
#pragma once
#include <cstdint>
#include <vector>

namespace zr {

struct BuildBox { double lo[3], hi[3]; };

struct BuildNode {
    BuildBox box;
    int32_t left = -1, right = -1;  // children (internal)
    uint32_t first = 0, count = 0;  // range in `order` (leaf when count > 0)
    uint32_t kind = 0;              // leaf kind (all objects of a leaf share it)
};

struct BuildResult {
    std::vector<BuildNode> nodes;  // nodes[0] = root
    std::vector<uint32_t> order;   // object ids, leaves reference contiguous ranges
    int max_depth = 0;             // depth of the deepest leaf, root = 0
};

// boxes/kinds: one per object.  Leaves hold <= max_leaf objects of one kind.  depth_limit bounds the
// depth of any leaf (the traversal stack is sized from it).
// max_leaf_kind[k] (may be null) caps the leaves of kind k below max_leaf: expensive objects with large boxes (cubes, media,
// wrapped objects) get leaves of their own.
void build_bvh(const std::vector<BuildBox>& boxes, const std::vector<uint32_t>& kinds, int max_leaf, int depth_limit,
               double cost_traverse, const double cost_kind[8], BuildResult& out, const int* max_leaf_kind = nullptr);

}  // namespace zr
