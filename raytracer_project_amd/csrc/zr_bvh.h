// zr_bvh.h — host-side BVH2 builder (binned SAH).  Replaces bvh_node's constructor
// (/root/reference/bvh.hpp:11-44: random axis, std::sort, median split, 26 s for 1M triangles) with a
// builder whose output is the flat sibling-pair array the kernels walk.  The tree may differ freely from
// the reference's: closest hit does not depend on it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <new>
#include <vector>

#include <sys/mman.h>

namespace zr {

struct BuildBox { double lo[3], hi[3]; };

struct BuildNode {
    BuildBox box;
    int32_t left, right;    // children (internal), -1 in a leaf
    uint32_t first, count;  // range in `order` (leaf when count > 0)
    uint32_t kind;          // leaf kind (all objects of a leaf share it)
};

// an array whose elements are NOT value-initialised by this process on allocation (the builder's threads touch the pages first:
// zero-filling 160 MB of nodes from one thread costs more than building the tree).  CONTRACT: allocate() hands out memory that
// READS AS ZERO until written — fresh anonymous mmap pages, which the kernel zero-fills on first touch.  The builder numbers its
// nodes sparsely and the flattener (zr_flatten.h: index_nodes, emit_run, run_pairs) scans whole id ranges, classifying slots the
// builder never wrote as "neither leaf nor inner" because they are zero.  Whoever changes the allocator (a pool, a reused
// mapping, malloc) must keep `zero_filled` true by clearing, or rewrite those scans as walks from the root.
template <class T>
class RawArray {
public:
    static constexpr bool zero_filled = true;   // see the contract above; asserted where it is relied on
    RawArray() = default;
    RawArray(const RawArray&) = delete;
    RawArray& operator=(const RawArray&) = delete;
    RawArray(RawArray&& o) noexcept : p_(o.p_), n_(o.n_), bytes_(o.bytes_) { o.p_ = nullptr; o.n_ = 0; o.bytes_ = 0; }
    RawArray& operator=(RawArray&& o) noexcept { if (this != &o) { clear(); p_ = o.p_; n_ = o.n_; bytes_ = o.bytes_; o.p_ = nullptr; o.n_ = 0; o.bytes_ = 0; } return *this; }
    ~RawArray() { clear(); }
    void allocate(size_t n) {
        clear();
        if (!n) return;
        bytes_ = (n * sizeof(T) + (2u << 20) - 1) / (2u << 20) * (2u << 20);
        void* q = ::mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) throw std::bad_alloc();
        (void)::madvise(q, bytes_, MADV_HUGEPAGE);   // a hint: 2 MB pages cut the first-touch faults of a 100 MB array 512-fold where the kernel grants them
        p_ = static_cast<T*>(q); n_ = n;
    }
    void shrink(size_t n) { if (n < n_) n_ = n; }
    void clear() { if (p_) ::munmap(p_, bytes_); p_ = nullptr; n_ = 0; bytes_ = 0; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T* data() { return p_; }
    const T* data() const { return p_; }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }
    const T* begin() const { return p_; }
    const T* end() const { return p_ + n_; }
private:
    T* p_ = nullptr; size_t n_ = 0, bytes_ = 0;
};
using NodeArray = RawArray<BuildNode>;

struct BuildResult {
    NodeArray nodes;               // nodes[0] = root
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
