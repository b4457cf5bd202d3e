// zr_host.cpp — implementation of the C ABI (include/zr_capi.h): scene ingest, BVH build, HBM upload,
// render driver.  Plain C++ (host compiler, -ffp-contract=off so that the camera frame and the sky
// constants are computed with exactly the reference's operation order, camera.hpp:358-399, 874-895,914);
// the kernels live in zr_kernels.hip.  There is deliberately no CPU rendering path in this library:
// without a HIP device zr_create() fails and says so.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <atomic>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/zr_capi.h"
#include "zr_build.h"
#include "zr_bvh.h"
#include "zr_device_types.h"
#include "zr_launch.h"

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_OK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(ZR_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

const double kInf = std::numeric_limits<double>::infinity();
const double kPi = 3.14159265358979323846;

struct H3 { double x, y, z; };
inline H3 h3(const double* p) { return H3{p[0], p[1], p[2]}; }
inline H3 operator+(H3 a, H3 b) { return H3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline H3 operator-(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 operator-(H3 a) { return H3{-a.x, -a.y, -a.z}; }
inline H3 operator*(double t, H3 v) { return H3{t * v.x, t * v.y, t * v.z}; }
inline H3 operator*(H3 v, double t) { return t * v; }
inline H3 operator/(H3 v, double t) { return (1 / t) * v; }  // vec3.hpp:149-151
inline H3 cross(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double len(H3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline H3 unit(H3 v) { double l = len(v); if (l < 1e-8) return H3{0, 0, 0}; return v / l; }
inline void st3(double* d, H3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (hi < v ? hi : v); }

template <class T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    int upload(const std::vector<T>& v) { return upload(v.data(), v.size()); }
    int upload(const zr::RawArray<T>& v) { return upload(v.data(), v.size()); }
    int upload(const T* src, size_t count) {
        release();
        n = count;
        size_t bytes = std::max<size_t>(sizeof(T) * count, 64);  // never a null device pointer
        HIP_OK(hipMalloc((void**)&p, bytes));
        if (count) HIP_OK(hipMemcpy(p, src, sizeof(T) * count, hipMemcpyHostToDevice));
        return ZR_OK;
    }
    int alloc(size_t count) {
        if (count == n && p) return ZR_OK;
        release();
        if (hipMalloc((void**)&p, std::max<size_t>(sizeof(T) * count, 64)) != hipSuccess) {
            p = nullptr; (void)hipGetLastError();
            return fail(ZR_E_DEVICE, "out of device memory (%zu bytes requested)", sizeof(T) * count);
        }
        n = count;
        return ZR_OK;
    }
};

}  // namespace

struct zr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t sub[ST_MAX_POOLS] = {};   // internal streams of the streaming pipeline's sub-pools 1..K-1 (sub[0] unused)
    DevBuf<unsigned long long> d_ctr;
    DevBuf<double> d_out;
    DevBuf<int32_t> d_tiles;
    bool warned_fallback = false;         // the notice about a frame beyond the streaming pipeline's limits has been printed
    int variant = 2;                      // 2 streaming wavefront pipeline (default); 0 pixel-group megakernel (the fallback for frames beyond the pipeline's packing limits)
    // variant 2: slot pool and per-frame buffers
    DevBuf<unsigned char> d_pool;
    DevBuf<uint32_t> d_pixels;
    DevBuf<double> d_partial;
    DevBuf<uint32_t> d_kend;               // reflection / refraction split: (draws, segments) of every unit's beauty path
    DevBuf<unsigned char> d_cls;           // ... and the class of its second path
    DevBuf<unsigned long long> d_cpart;    // ... and the per-block counters of the two passes
    DevBuf<unsigned int> d_ctl;
    DevBuf<unsigned char> d_st_overflow;
    uint32_t st_ovf_levels = 0;           // levels per lane the spill slabs of d_st_overflow hold
    int st_blocks = 0;
    int fused_blocks = 0;                 // persistent grid of the fused small-scene kernel (0: not asked yet)
    uint32_t st_slots = 0;
    int st_pools = -1;                    // sub-pools staggered on separate streams: 1 = one pool, -1 = auto
    hipEvent_t st_event = nullptr;
    unsigned int* h_active = nullptr;     // pinned
    std::vector<int32_t> pix_key;         // plan the cached pixel list was built for
    uint64_t last_rounds = 0;
    int last_path = 0;                    // zr_counters::path of the last render
    double last_extend_ms = 0, last_shade_ms = 0;
    // device timing of render-kernel launches: HIP events recorded on the stream the kernel is launched on
    struct Pending { hipEvent_t a, b; uint64_t render_id; int kind; };
    std::vector<hipEvent_t> pool;     // recycled events
    std::vector<Pending> pending;     // launches not yet resolved to milliseconds
    std::vector<float> log;           // resolved launch times since the last zr_get_kernel_times (oldest first)
    int log_kind = 1;                 // ZR_TIMELOG_KIND: which kernel zr_get_kernel_times reports (1 extend/render, 2 shade)
    uint64_t render_id = 0;
    double last_render_ms = 0;        // sum over the launches of the most recent render call
    hipStream_t last_stream = nullptr;
    bool last_counted = false;
};

// one input array of a scene: the library's own copy (zr_scene_set_*) or a view of the caller's memory (zr_scene_set_all_borrowed)
template <class T>
struct HostArray {
    const T* p = nullptr; size_t n = 0;
    std::vector<T> own;
    void copy(const T* src, size_t count) { own.assign(src, src + count); p = own.data(); n = count; }
    void borrow(const T* src, size_t count) { std::vector<T>().swap(own); p = src; n = count; }
    void drop() { std::vector<T>().swap(own); p = nullptr; n = 0; }
    const T& operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const T* data() const { return p; }
    const T* begin() const { return p; }
    const T* end() const { return p + n; }
};

struct zr_scene {
    zr_ctx* ctx = nullptr;
    // host side of the world as given: copies, or borrowed views until the commit
    HostArray<double> spheres, tri_v, tri_n, cubes;
    HostArray<uint32_t> sphere_mat, tri_mat, cube_mat;
    HostArray<zr_medium> media;
    HostArray<zr_xform_op> ops;
    HostArray<zr_object> objects;
    bool objects_set = false;
    std::vector<zr_group> groups;   // runs of triangles that ZR_PRIM_GROUP objects place (small: copied)
    bool borrowed = false;        // the geometry arrays are the caller's (released after the commit)
    bool released = false;        // ... and have been released: the scene cannot be committed again without new input
    unsigned reset_mask = 0;      // ... which geometry arrays have been given again since (SET_* bits, zr_host.cpp: GEOMETRY_SET)
    std::vector<zr_material> materials;
    std::vector<zr_texture> textures;
    HostArray<unsigned char> texels;
    // device
    bool committed = false;
    DevBuf<zr::NodePair> d_nodes;
    DevBuf<zr::NodeQ> d_quads;
    bool quad_ok = true;          // leaf references fit the EXTEND kernel's 32-bit stack entries
    DevBuf<double> d_spheres, d_tri_v, d_tri_s, d_cubes, d_pcubes;
    DevBuf<uint32_t> d_sphere_mat, d_cube_mat, d_pcube_mat;
    DevBuf<zr::DMedium> d_media;
    DevBuf<zr::DWrapped> d_wrapped;
    DevBuf<zr::DInstance> d_insts;
    DevBuf<zr_xform_op> d_ops;
    DevBuf<zr_material> d_mats;
    DevBuf<zr_texture> d_texs;
    DevBuf<unsigned char> d_texels;
    zr::DScene ds{};
    int leaf_level = 2;           // EXTEND build: 0 bare triangles / spheres only, 1 + bare and placed cubes and unwrapped media, 2 everything
    uint32_t stack_demand = 0;    // worst-case entries on an EXTEND lane's traversal stack (Flattener::stack_demand)
    size_t leaf_objects = 0;      // leaf objects of the world's tree (all kinds): small worlds render through the fused kernel
    zr::FusedObjs fused{};        // ... their records, for the kernel arguments (finish_commit)
    bool fused_ok = false;
    uint64_t stats[4] = {0, 0, 0, 0};
    const char* builder = "";   // which builder made the committed tree (zr_scene_builder)
};

namespace {

// ---- bounding boxes of world-list entries, following the reference's constructors --------------------
struct Boxer {
    const zr_scene& s;
    const std::vector<zr::BuildBox>* group_box = nullptr;   // per zr_group: the box of its triangles in their own space
    zr::BuildBox prim(uint32_t type, uint32_t idx) const {
        zr::BuildBox b;
        if (type == ZR_PRIM_GROUP) return (*group_box)[idx];
        if (type == ZR_PRIM_SPHERE) {  // sphere.hpp:12-14 (raw radius argument)
            const double* q = &s.spheres[(size_t)idx * 4];
            for (int k = 0; k < 3; k++) { b.lo[k] = std::fmin(q[k] - q[3], q[k] + q[3]); b.hi[k] = std::fmax(q[k] - q[3], q[k] + q[3]); }
        } else if (type == ZR_PRIM_TRIANGLE) {  // triangle.hpp:84-101
            const double* v = &s.tri_v[(size_t)idx * 9];
            for (int k = 0; k < 3; k++) {
                b.lo[k] = std::fmin(v[k], std::fmin(v[3 + k], v[6 + k]));
                b.hi[k] = std::fmax(v[k], std::fmax(v[3 + k], v[6 + k]));
                if (b.hi[k] - b.lo[k] < 0.0001) { b.lo[k] -= 0.0001; b.hi[k] += 0.0001; }
            }
        } else if (type == ZR_PRIM_CUBE) {  // cube.hpp:34-41
            const double* q = &s.cubes[(size_t)idx * 12];
            for (int k = 0; k < 3; k++) { b.lo[k] = q[6 + k] - 0.00005; b.hi[k] = q[9 + k] + 0.00005; }
        } else {  // constant_medium.hpp:79-81
            const zr_medium& m = s.media[idx];
            b = chain(m.boundary_type, m.boundary_index, m.chain_first, m.chain_count);
        }
        return b;
    }
    zr::BuildBox chain(uint32_t type, uint32_t idx, uint32_t cf, uint32_t cn) const {
        if (cn == 0) return prim(type, idx);
        zr::BuildBox in = chain(type, idx, cf + 1, cn - 1), b;
        const zr_xform_op& op = s.ops[cf];
        if (op.kind == ZR_OP_TRANSLATE) {  // translate.hpp:12
            for (int k = 0; k < 3; k++) { b.lo[k] = in.lo[k] + op.a[k]; b.hi[k] = in.hi[k] + op.a[k]; }
            return b;
        }
        if (op.kind == ZR_OP_SCALE) {  // scale.hpp:11-17
            for (int k = 0; k < 3; k++) { double a0 = in.lo[k] * op.a[k], a1 = in.hi[k] * op.a[k]; b.lo[k] = std::fmin(a0, a1); b.hi[k] = std::fmax(a0, a1); }
            return b;
        }
        if (op.kind == ZR_OP_MATERIAL) return in;
        for (int k = 0; k < 3; k++) { b.lo[k] = kInf; b.hi[k] = -kInf; }
        const double sn = op.a[0], co = op.a[1];  // rotate_*.hpp constructors: the 8 corners
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
            double x = i ? in.hi[0] : in.lo[0], y = j ? in.hi[1] : in.lo[1], z = k ? in.hi[2] : in.lo[2];
            double t[3] = {x, y, z};
            // rotate_y.hpp:26-27 builds its box with the INVERSE rotation (+sin) although hit() maps object points with
            // (cos x - sin z, sin x + cos z) (rotate_y.hpp:63-64): for children that are not symmetric about the y axis the
            // reference's box misses real geometry and what gets culled depends on its random tree.  Bound the true geometry.
            if (op.kind == ZR_OP_ROTATE_Y) { t[0] = co * x - sn * z; t[2] = sn * x + co * z; }
            else if (op.kind == ZR_OP_ROTATE_X) { t[1] = co * y - sn * z; t[2] = sn * y + co * z; }
            else { t[0] = co * x - sn * y; t[1] = sn * x + co * y; }
            for (int q = 0; q < 3; q++) { b.lo[q] = std::fmin(b.lo[q], t[q]); b.hi[q] = std::fmax(b.hi[q], t[q]); }
        }
        return b;
    }
};

inline float f_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return std::nextafterf(f, -std::numeric_limits<float>::infinity());
}
inline float f_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return std::nextafterf(f, std::numeric_limits<float>::infinity());
}

// what a world-list entry becomes in the tree: its leaf kind, and whether the host stores it "baked" (0 as is, 1 baked triangle,
// 2 material-only chain, 3 baked sphere, 4 placed cube) — see Flattener::put_baked_triangle / put_baked_sphere / put_pcube
inline void classify_object(const zr_scene& s, const zr_object& o, bool bake, uint32_t& kind, uint8_t& baked) {
    kind = o.chain_count ? ZR_KIND_WRAPPED : o.type;
    baked = 0;
    if (o.type == ZR_PRIM_GROUP) { kind = ZR_KIND_INSTANCE; return; }   // placed as one object, whatever its chain
    if (!bake || o.chain_count == 0) return;
    if (o.type == ZR_PRIM_TRIANGLE) {   // see Flattener::put_baked_triangle
        bool ok = true;
        for (uint32_t q = 0; q < o.chain_count; q++) if (s.ops[o.chain_first + q].kind == ZR_OP_SCALE) ok = false;
        if (ok) { baked = 1; kind = ZR_PRIM_TRIANGLE; }
    }
    if (o.type == ZR_PRIM_SPHERE) {   // see Flattener::put_baked_sphere
        bool ok = true, moved = false; uint32_t mat = s.sphere_mat[o.index];
        for (int q = (int)o.chain_count - 1; q >= 0 && ok; q--) {
            const zr_xform_op& op = s.ops[o.chain_first + q];
            if (op.kind == ZR_OP_SCALE) { ok = op.a[0] > 0 && op.a[0] == op.a[1] && op.a[1] == op.a[2]; moved = true; }
            else if (op.kind == ZR_OP_TRANSLATE) moved = true;
            else if (op.kind == ZR_OP_MATERIAL) mat = op.mat;
            else ok = false;
        }
        if (ok && moved && mat < 0x7FFFFFFFu) { baked = 3; kind = ZR_PRIM_SPHERE; }
    }
    if (o.type == ZR_PRIM_CUBE) {   // see Flattener::put_pcube: [translate], [translate, rotate_y], each optionally followed by a scale — outermost first
        int pat = 0; bool ok = true;   // 0 nothing yet, 1 translate seen, 2 translate then rotate_y seen, 3 ... then a scale (the innermost wrapper)
        for (uint32_t q = 0; q < o.chain_count && ok; q++) {
            const zr_xform_op& op = s.ops[o.chain_first + q];
            const uint32_t kd = op.kind;
            if (kd == ZR_OP_MATERIAL) continue;
            if (kd == ZR_OP_TRANSLATE && pat == 0) pat = 1;
            else if (kd == ZR_OP_ROTATE_Y && pat == 1) pat = 2;
            else if (kd == ZR_OP_SCALE && (pat == 1 || pat == 2) && op.a[0] != 0.0 && op.a[1] != 0.0 && op.a[2] != 0.0) pat = 3;
            else ok = false;
        }
        if (ok && pat >= 1) { baked = 4; kind = ZR_KIND_PCUBE; }
    }
    if (!baked && (o.type == ZR_PRIM_SPHERE || o.type == ZR_PRIM_CUBE)) {
        bool only_material = true;
        for (uint32_t q = 0; q < o.chain_count; q++) if (s.ops[o.chain_first + q].kind != ZR_OP_MATERIAL) only_material = false;
        if (only_material) { baked = 2; kind = o.type; }
    }
}

int validate(const zr_scene& s, const std::vector<zr_object>& objs) {
    const size_t nm = s.materials.size(), nt = s.textures.size();
    auto mat_ok = [&](uint32_t m) { return m == 0xFFFFFFFFu || m < nm; };
    for (uint32_t m : s.sphere_mat) if (!mat_ok(m)) return fail(ZR_E_INVALID, "sphere material id %u out of range", m);
    for (uint32_t m : s.tri_mat) if (!mat_ok(m)) return fail(ZR_E_INVALID, "triangle material id %u out of range", m);
    for (uint32_t m : s.cube_mat) if (!mat_ok(m)) return fail(ZR_E_INVALID, "cube material id %u out of range", m);
    auto chain_ok = [&](uint32_t cf, uint32_t cn) {
        if (cn > ZR_MAX_CHAIN || (size_t)cf + cn > s.ops.size()) return false;
        for (uint32_t k = 0; k < cn; k++) {
            const zr_xform_op& op = s.ops[cf + k];
            if (op.kind > ZR_OP_MATERIAL) return false;
            if (op.kind == ZR_OP_MATERIAL && !mat_ok(op.mat)) return false;
        }
        return true;
    };
    auto prim_ok = [&](uint32_t type, uint32_t idx) {
        switch (type) {
            case ZR_PRIM_SPHERE: return idx < s.sphere_mat.size();
            case ZR_PRIM_TRIANGLE: return idx < s.tri_mat.size();
            case ZR_PRIM_CUBE: return idx < s.cube_mat.size();
            case ZR_PRIM_MEDIUM: return idx < s.media.size();
            case ZR_PRIM_GROUP: return idx < s.groups.size();
            default: return false;
        }
    };
    for (const zr_medium& m : s.media) {
        if (m.boundary_type != ZR_PRIM_SPHERE && m.boundary_type != ZR_PRIM_CUBE) return fail(ZR_E_INVALID, "medium boundary must be a sphere or a cube");
        if (!prim_ok(m.boundary_type, m.boundary_index) || !chain_ok(m.chain_first, m.chain_count) || !mat_ok(m.mat))
            return fail(ZR_E_INVALID, "medium references out of range (or wrapper chain longer than %d)", ZR_MAX_CHAIN);
    }
    for (const zr_group& g : s.groups)
        if (g.triangle_count == 0 || (size_t)g.first_triangle + g.triangle_count > s.tri_mat.size()) return fail(ZR_E_INVALID, "group of triangles out of range (or empty)");
    if (!s.groups.empty() && !s.objects_set) return fail(ZR_E_INVALID, "groups need an explicit world list (zr_scene_set_objects)");
    for (const zr_object& o : objs)
        if (!prim_ok(o.type, o.index) || !chain_ok(o.chain_first, o.chain_count))
            return fail(ZR_E_INVALID, "world-list entry references out of range (or wrapper chain longer than %d)", ZR_MAX_CHAIN);
    for (const zr_material& m : s.materials) {
        if (m.kind > ZR_MAT_ISOTROPIC) return fail(ZR_E_INVALID, "unknown material kind %u", m.kind);
        if (m.kind != ZR_MAT_DIELECTRIC && m.tex >= nt) return fail(ZR_E_INVALID, "material texture id out of range");
        if (m.bump_tex != ZR_NO_TEXTURE && m.bump_tex >= nt) return fail(ZR_E_INVALID, "material bump texture id out of range");
    }
    for (const zr_texture& t : s.textures) {
        if (t.kind > ZR_TEX_IMAGE_F32) return fail(ZR_E_INVALID, "unknown texture kind %u", t.kind);
        if (t.kind == ZR_TEX_CHECKER && (t.odd >= nt || t.even >= nt)) return fail(ZR_E_INVALID, "checker child texture out of range");
        if (t.kind >= ZR_TEX_IMAGE_U8 && t.width && t.height) {
            size_t bytes = (size_t)t.width * t.height * 3 * (t.kind == ZR_TEX_IMAGE_F32 ? 4 : 1);
            if (t.texel_offset + bytes > s.texels.size()) return fail(ZR_E_INVALID, "image texture texels out of range");
            if (t.kind == ZR_TEX_IMAGE_F32 && (t.texel_offset & 3)) return fail(ZR_E_INVALID, "float texels must be 4-byte aligned");
        }
    }
    return ZR_OK;
}

// flattens the build tree into sibling-pair records and 4-wide nodes and, leaf by leaf, the primitive arrays in leaf order.
// Built for commit latency like the builder (zr_bvh.cpp): one cheap serial walk fixes every index (pair numbers in pre-order,
// each leaf's range in its kind's array), then the primitive records, the pair records and the 4-wide nodes (whose shape
// depends on quantisation trials) are produced by all threads; nothing is appended under a lock.  The arrays are
// zr::RawArray (no zero-fill).  The result is the same as a serial depth-first emit, whatever the number of threads.
double env_double(const char* name, double dflt);
struct Flattener {
    const zr_scene& s;
    const std::vector<zr_object>& objs;
    const zr::BuildResult& br;
    zr::RawArray<zr::NodePair> pairs;
    std::vector<zr::NodeQ> quads;
    zr::RawArray<uint32_t> leaf_first; // per build node: device index of a leaf's first primitive
    int quad_depth = 0;
    zr::RawArray<double> spheres, tri_v, tri_s, cubes, pcubes;
    zr::RawArray<uint32_t> sphere_mat, cube_mat, pcube_mat;
    zr::RawArray<zr::DMedium> media;
    zr::RawArray<zr::DWrapped> wrapped;
    zr::RawArray<zr::DInstance> insts;
    std::vector<uint32_t> inst_group;                     // per placement: its group
    const std::vector<zr::BuildResult>* runs = nullptr;   // per zr_group: the tree over its triangles (object space), built by the caller
    std::vector<uint32_t> run_root;                       // per group: pair index of its subtree's root
    std::vector<uint32_t> run_tri_base, run_qroot, run_demand;   // per group: first triangle (device index), its root among the 4-wide nodes, its worst-case stack entries
    bool root_in_array = false;                           // a group's own flattener: the root is a quantised node like any other (quads[0])
    size_t n_sph = 0, n_tri = 0, n_cube = 0, n_pcube = 0, n_media = 0, n_wrapped = 0;   // filled sizes (the arrays are sized exactly)
    std::function<void()> after_primitives;   // called by run() once spheres / triangles / cubes / media / wrapped are complete
    const std::vector<uint8_t>* baked = nullptr;   // per object: 0 as is, 1 baked triangle, 2 material-only chain, 3 baked sphere, 4 placed cube
    size_t n_baked = 0;
    int threads = 1;

    // Workers that live as long as run(): the level-synchronous passes below call parallel_for some forty times, and starting
    // thirty-one threads each time cost more than the passes' own work (plan + numbering 34 ms -> see profiles/r2_commit_stats.txt).
    struct Pool {
        std::vector<std::thread> th;
        std::mutex m;
        std::condition_variable go, done;
        std::function<void(int)> job;
        uint64_t gen = 0;
        int parts = 0, pending = 0;
        bool stop = false;
        explicit Pool(int workers) {
            for (int w = 1; w <= workers; w++)
                th.emplace_back([this, w]() {
                    uint64_t seen = 0;
                    for (;;) {
                        std::function<void(int)> f;
                        {
                            std::unique_lock<std::mutex> lk(m);
                            go.wait(lk, [&] { return stop || gen != seen; });
                            if (stop) return;
                            seen = gen;
                            if (w >= parts) continue;
                            f = job;
                        }
                        f(w);
                        { std::lock_guard<std::mutex> lk(m); if (--pending == 0) done.notify_one(); }
                    }
                });
        }
        ~Pool() {
            { std::lock_guard<std::mutex> lk(m); stop = true; }
            go.notify_all();
            for (auto& x : th) x.join();
        }
        void run(int n_parts, const std::function<void(int)>& f) {   // f(0 .. n_parts - 1), part 0 on the caller
            { std::lock_guard<std::mutex> lk(m); job = f; parts = n_parts; pending = n_parts - 1; gen++; }
            go.notify_all();
            f(0);
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return pending == 0; });
        }
    };
    mutable std::unique_ptr<Pool> pool;
    template <class F>
    void parallel_for(size_t n, size_t grain, F&& fn) const {   // fn(begin, end) over [0, n) split evenly
        int T = threads;
        if (n < 2 * grain) T = 1; else T = (int)std::min<size_t>((size_t)T, n / grain);
        if (T <= 1) { fn((size_t)0, n); return; }
        if (pool && T <= (int)pool->th.size() + 1) {
            pool->run(T, [&fn, n, T](int t) { fn(n * (size_t)t / (size_t)T, n * (size_t)(t + 1) / (size_t)T); });
            return;
        }
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back([&fn, n, t, T]() { fn(n * t / T, n * (t + 1) / T); });
        fn((size_t)0, n / T);
        for (auto& x : th) x.join();
    }

    // ---- one primitive record at a given index of its kind's array ------------------------------------------------------
    static constexpr uint32_t kKeepMaterial = 0xFFFFFFFEu;
    // `mat` != kKeepMaterial: the primitive sits under material_instance wrappers only, which do nothing but replace rec.mat
    // (material_instance.hpp:12-28) — it is stored bare with the outermost instance's material
    void put_sphere(size_t di, uint32_t idx, uint32_t mat) {
        const double* q = &s.spheres[(size_t)idx * 4];
        double* d = &spheres[di * 4];
        d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = std::fmax(0, q[3]);  // sphere.hpp:9
        sphere_mat[di] = mat != kKeepMaterial ? mat : s.sphere_mat[idx];
    }
    void put_triangle_raw(size_t di, const double* v, const double* nn, uint32_t mat, bool force_front) {
        std::memcpy(&tri_v[di * ZR_TRI_STRIDE], v, 72);
        double* t = &tri_s[di * 20];
        std::memcpy(t, v, 72); std::memcpy(t + 9, nn, 72);
        uint64_t mbits = mat, fbits = force_front ? 1u : 0u;
        std::memcpy(t + 18, &mbits, 8); std::memcpy(t + 19, &fbits, 8);
    }
    void put_triangle(size_t di, uint32_t idx) { put_triangle_raw(di, &s.tri_v[(size_t)idx * 9], &s.tri_n[(size_t)idx * 9], s.tri_mat[idx], false); }
    void put_cube(size_t di, uint32_t idx, uint32_t mat) {
        std::memcpy(&cubes[di * 6], &s.cubes[(size_t)idx * 12], 48);
        cube_mat[di] = mat != kKeepMaterial ? mat : s.cube_mat[idx];
    }
    // A triangle under a chain of translate / rotate_x,y,z / material_instance wrappers is stored in WORLD space as a bare
    // triangle: vertices and (un-normalised) vertex normals mapped object -> world with the wrappers' own forward maps
    // (translate.hpp:24-27, rotate_*.hpp hit(): the maps apply_op_rec uses for hit points), material = the outermost
    // material_instance, plus a flag when the chain holds a translate or a rotate_y, which force front_face = true
    // (SURVEY §8 a-17 quirk).  t is the same in both spaces (the wrappers do not normalise the transformed direction), so
    // only the last bits of the hit differ from transforming the ray — and every mesh the reference's scenes place in the
    // world (model -> material_instance -> rotate -> translate) runs on the bare-triangle fast path instead of paying a
    // chain transform per candidate.  scale is excluded: it would change which triangles count as degenerate.
    void put_baked_triangle(size_t di, const zr_object& o) {
        double v[9], nn[9];
        std::memcpy(v, &s.tri_v[(size_t)o.index * 9], sizeof v);
        std::memcpy(nn, &s.tri_n[(size_t)o.index * 9], sizeof nn);
        uint32_t mat = s.tri_mat[o.index];
        bool force_front = false;
        for (int k = (int)o.chain_count - 1; k >= 0; k--) {   // innermost wrapper first, as the hit record travels outwards
            const zr_xform_op& op = s.ops[o.chain_first + k];
            const double sn = op.a[0], co = op.a[1];
            for (int c = 0; c < 3; c++) {
                double* p = v + 3 * c; double* q = nn + 3 * c;
                switch (op.kind) {
                    case ZR_OP_TRANSLATE: p[0] += op.a[0]; p[1] += op.a[1]; p[2] += op.a[2]; break;
                    case ZR_OP_ROTATE_Y: { double x = p[0], z = p[2]; p[0] = co * x - sn * z; p[2] = sn * x + co * z;
                                           x = q[0]; z = q[2]; q[0] = co * x - sn * z; q[2] = sn * x + co * z; } break;
                    case ZR_OP_ROTATE_X: { double y = p[1], z = p[2]; p[1] = co * y - sn * z; p[2] = sn * y + co * z;
                                           y = q[1]; z = q[2]; q[1] = co * y - sn * z; q[2] = sn * y + co * z; } break;
                    case ZR_OP_ROTATE_Z: { double x = p[0], y = p[1]; p[0] = co * x - sn * y; p[1] = sn * x + co * y;
                                           x = q[0]; y = q[1]; q[0] = co * x - sn * y; q[1] = sn * x + co * y; } break;
                    default: break;
                }
            }
            if (op.kind == ZR_OP_TRANSLATE || op.kind == ZR_OP_ROTATE_Y) force_front = true;
            if (op.kind == ZR_OP_MATERIAL) mat = op.mat;
        }
        put_triangle_raw(di, v, nn, mat, force_front);
    }
    // A sphere under uniform scale / translate / material_instance wrappers (the demo scene's instanced spheres:
    // scale -> material_instance -> translate) is the sphere (c s + offset, r s): same t, same unit normal, same u/v and
    // tangent (no rotation involved); bit 31 of its material word records the front_face = true a translate forces.
    void put_baked_sphere(size_t di, const zr_object& o) {
        const double* q = &s.spheres[(size_t)o.index * 4];
        double c[3] = {q[0], q[1], q[2]}, r = std::fmax(0, q[3]);
        uint32_t mat = s.sphere_mat[o.index];
        bool force_front = false;
        for (int k = (int)o.chain_count - 1; k >= 0; k--) {
            const zr_xform_op& op = s.ops[o.chain_first + k];
            if (op.kind == ZR_OP_SCALE) { for (double& x : c) x *= op.a[0]; r *= op.a[0]; }
            else if (op.kind == ZR_OP_TRANSLATE) { c[0] += op.a[0]; c[1] += op.a[1]; c[2] += op.a[2]; force_front = true; }
            else if (op.kind == ZR_OP_MATERIAL) mat = op.mat;
        }
        double* d = &spheres[di * 4];
        d[0] = c[0]; d[1] = c[1]; d[2] = c[2]; d[3] = r;
        sphere_mat[di] = force_front ? (mat | 0x80000000u) : mat;
    }
    // A cube under translate, or under rotate_y then translate, either with a scale as the innermost wrapper (material_instance wrappers
    // anywhere) — how every cube of the reference's scenes is placed (scene_management.hpp:132-139 and the scaled, turned instances of its
    // master cube, :178-201; cfg5's walls and boxes) — is stored as a PLACED CUBE: the cube's own numbers plus the wrappers' parameters in one
    // 128-byte record.  The device applies the wrappers' ray and hit-record maps in the
    // chain's order with the chain's arithmetic (zr_device.h pcube_ray / object_rec), so results are those of the wrapped object;
    // what is saved is the op-list loop, its loads and the registers of the generic chain code in the traversal kernel.
    void put_pcube(size_t di, const zr_object& o) {
        const double* q = &s.cubes[(size_t)o.index * 12];
        double rec[ZR_PCUBE_STRIDE] = {q[0], q[1], q[2], q[3], q[4], q[5], 0, 0, 0, 0, 1, 0, 1, 1, 1, 0};
        uint32_t mat = s.cube_mat[o.index];
        for (int k = (int)o.chain_count - 1; k >= 0; k--) {   // inside-out: the outermost material_instance is applied last
            const zr_xform_op& op = s.ops[o.chain_first + k];
            if (op.kind == ZR_OP_TRANSLATE) { rec[6] = op.a[0]; rec[7] = op.a[1]; rec[8] = op.a[2]; }
            else if (op.kind == ZR_OP_ROTATE_Y) { rec[9] = op.a[0]; rec[10] = op.a[1]; rec[11] = 1.0; }
            else if (op.kind == ZR_OP_SCALE) { rec[12] = op.a[0]; rec[13] = op.a[1]; rec[14] = op.a[2]; rec[15] = 1.0; }
            else if (op.kind == ZR_OP_MATERIAL) mat = op.mat;
        }
        std::memcpy(&pcubes[di * ZR_PCUBE_STRIDE], rec, sizeof rec);
        pcube_mat[di] = mat;
    }
    // object `oi` as a leaf primitive of a plain kind (sphere / triangle / cube / placed cube) at index di of that kind's array
    void put_leaf_object(uint32_t oi, size_t di) {
        const zr_object& o = objs[oi];
        const uint8_t bk = baked ? (*baked)[oi] : 0;
        if (o.type == ZR_PRIM_GROUP) { zr::DInstance in{}; in.chain_first = o.chain_first; in.chain_count = o.chain_count; in.root = run_root[o.index]; insts[di] = in; inst_group[di] = o.index; }
        else if (bk == 1) put_baked_triangle(di, o);
        else if (bk == 3) put_baked_sphere(di, o);
        else if (bk == 4) put_pcube(di, o);
        else {
            const uint32_t mat = bk == 2 ? s.ops[o.chain_first].mat : kKeepMaterial;   // material-only chain: the outermost wrapper is applied last
            if (o.type == ZR_PRIM_SPHERE) put_sphere(di, o.index, mat);
            else if (o.type == ZR_PRIM_TRIANGLE) put_triangle(di, o.index);
            else put_cube(di, o.index, mat);
        }
    }
    // a primitive that is not a leaf object itself — a medium's boundary, the object inside a wrapper chain — goes behind the leaf
    // ranges of its kind's array (serial: such objects are few)
    uint32_t append_inner(uint32_t type, uint32_t idx) {
        switch (type) {
            case ZR_PRIM_SPHERE: put_sphere(n_sph, idx, kKeepMaterial); return (uint32_t)n_sph++;
            case ZR_PRIM_TRIANGLE: put_triangle(n_tri, idx); return (uint32_t)n_tri++;
            case ZR_PRIM_CUBE: put_cube(n_cube, idx, kKeepMaterial); return (uint32_t)n_cube++;
            default: { const size_t di = n_media++; put_medium(di, idx); return (uint32_t)di; }
        }
    }
    // A group's subtree: its triangles, unbaked, behind the leaf ranges of the triangle arrays in the order its leaves name them, and
    // its sibling-pair records from pair index `base` on, numbered in pre-order (= ascending build-node id, as in index_nodes).
    // Returns the number of pair records written (at least one: a run that fits one leaf gets a pair with an empty second child).
    uint32_t emit_run(uint32_t g, uint32_t base) {
        const zr::BuildResult& rb = (*runs)[g];
        static_assert(zr::NodeArray::zero_filled, "the id-range scans below read slots the builder never wrote: they must read as zero");
        const zr_group& grp = s.groups[g];
        // the run's triangles in the order of its leaves by ascending node id — the order index_nodes() gives a tree's leaves, so that
        // the run's 4-wide nodes (emit_run_quads, numbered by a flattener of their own) name the same indices
        run_tri_base[g] = (uint32_t)n_tri;
        std::vector<uint32_t> first_of(rb.nodes.size(), 0);
        for (size_t id = 0; id < rb.nodes.size(); id++) {
            const zr::BuildNode& n = rb.nodes[id];
            if (!n.count) continue;
            first_of[id] = (uint32_t)n_tri;
            for (uint32_t k = 0; k < n.count; k++) { put_triangle(n_tri, grp.first_triangle + rb.order[n.first + k]); n_tri++; }
        }
        auto leaf_tris = [&](const zr::BuildNode& n) { return first_of[(size_t)(&n - &rb.nodes[0])]; };
        if (rb.nodes[0].count) {
            const zr::BuildNode& n = rb.nodes[0];
            for (int k = 0; k < 3; k++) { pairs[base].lo[0][k] = f_down(n.box.lo[k]); pairs[base].hi[0][k] = f_up(n.box.hi[k]); }
            pairs[base].child[0] = leaf_tris(n); pairs[base].meta[0] = ((ZR_PRIM_TRIANGLE + 1u) << 16) | n.count;
            empty_child(base, 1);
            return 1;
        }
        std::vector<uint32_t> ids;   // inner nodes, ascending id = pre-order
        for (size_t id = 0; id < rb.nodes.size(); id++) { const zr::BuildNode& n = rb.nodes[id]; if (n.count == 0 && n.left == (int32_t)id + 1) ids.push_back((uint32_t)id); }
        auto pair_index = [&](uint32_t id) { return base + (uint32_t)(std::lower_bound(ids.begin(), ids.end(), id) - ids.begin()); };
        for (size_t p = 0; p < ids.size(); p++) {
            const zr::BuildNode& n = rb.nodes[ids[p]];
            const int32_t ch[2] = {n.left, n.right};
            for (int slot = 0; slot < 2; slot++) {
                const zr::BuildNode& c = rb.nodes[ch[slot]];
                zr::NodePair& pr = pairs[base + p];
                for (int k = 0; k < 3; k++) { pr.lo[slot][k] = f_down(c.box.lo[k]); pr.hi[slot][k] = f_up(c.box.hi[k]); }
                if (c.count) { pr.child[slot] = leaf_tris(c); pr.meta[slot] = ((ZR_PRIM_TRIANGLE + 1u) << 16) | c.count; }
                else { pr.child[slot] = pair_index((uint32_t)ch[slot]); pr.meta[slot] = 0; }
            }
        }
        return (uint32_t)ids.size();
    }
    static uint32_t run_pairs(const zr::BuildResult& rb) {   // pair records emit_run will write
        if (rb.nodes.empty() || rb.nodes[0].count) return 1;
        uint32_t n = 0;
        for (size_t id = 0; id < rb.nodes.size(); id++) { const zr::BuildNode& q = rb.nodes[id]; if (q.count == 0 && q.left == (int32_t)id + 1) n++; }
        return n;
    }
    void put_medium(size_t di, uint32_t idx) {
        const zr_medium& m = s.media[idx];
        zr::DMedium d{};
        d.btype = m.boundary_type; d.chain_first = m.chain_first; d.chain_count = m.chain_count;
        d.mat = m.mat; d.id = idx; d.neg_inv_density = m.neg_inv_density;
        d.bindex = append_inner(m.boundary_type, m.boundary_index);
        media[di] = d;
    }

    // ---- the serial walk: pair numbers in pre-order, leaf ranges per kind, in the order a depth-first emit would visit them ----
    std::vector<int32_t> inner;        // inner build nodes, position = pair index
    zr::RawArray<uint32_t> pair_of;    // per build node
    std::vector<int32_t> leaves;       // leaf build nodes in emit order
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // leaf objects per kind
    // Pair index of every inner node, first-primitive index of every leaf, and the two lists, in the order of a depth-first walk
    // (node, left subtree, right subtree).  The builder numbers nodes so that this order IS ascending node id (left = id + 1,
    // right = id + 2 x the left subtree's references: zr_bvh.cpp), with unused ids in between — fresh pages, all zero, which no
    // real node is (a leaf has count > 0, an inner node left = id + 1 > 0) — so the walk is two passes of prefix sums over the
    // id range, every thread on its own slice, instead of a serial recursion over two million nodes.
    void index_nodes() {
        static_assert(zr::NodeArray::zero_filled, "the unused ids between real nodes must read as zero (neither leaf nor inner)");
        const size_t N = br.nodes.size();
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), N / 65536 + 1));
        struct Tally { size_t inner = 0, leaves = 0; uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
        std::vector<Tally> tally((size_t)T);
        auto is_inner = [&](size_t id) { const zr::BuildNode& n = br.nodes[id]; return n.count == 0 && n.left == (int32_t)id + 1; };
        auto pass = [&](auto&& body) {
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) th.emplace_back([&body, t, T, N]() { body(t, N * (size_t)t / (size_t)T, N * (size_t)(t + 1) / (size_t)T); });
            body(0, (size_t)0, N / (size_t)T);
            for (auto& x : th) x.join();
        };
        pass([&](int t, size_t a, size_t b) {
            Tally y;
            for (size_t id = a; id < b; id++) {
                const zr::BuildNode& n = br.nodes[id];
                if (n.count) { y.leaves++; y.cnt[n.kind & 7] += n.count; } else if (is_inner(id)) y.inner++;
            }
            tally[(size_t)t] = y;
        });
        Tally run;
        std::vector<Tally> start((size_t)T);
        for (int t = 0; t < T; t++) {
            start[(size_t)t] = run;
            run.inner += tally[(size_t)t].inner; run.leaves += tally[(size_t)t].leaves;
            for (int k = 0; k < 8; k++) run.cnt[k] += tally[(size_t)t].cnt[k];
        }
        inner.resize(run.inner); leaves.resize(run.leaves);
        for (int k = 0; k < 8; k++) cnt[k] = run.cnt[k];
        pass([&](int t, size_t a, size_t b) {
            Tally y = start[(size_t)t];
            for (size_t id = a; id < b; id++) {
                const zr::BuildNode& n = br.nodes[id];
                if (n.count) { leaf_first[id] = y.cnt[n.kind & 7]; y.cnt[n.kind & 7] += n.count; leaves[y.leaves++] = (int32_t)id; }
                else if (is_inner(id)) { pair_of[id] = (uint32_t)y.inner; inner[y.inner++] = (int32_t)id; }
            }
        });
    }
    void fill_pair(uint32_t p, int32_t node_id) {
        const int32_t ch[2] = {br.nodes[node_id].left, br.nodes[node_id].right};
        for (int slot = 0; slot < 2; slot++) {
            const zr::BuildNode& n = br.nodes[ch[slot]];
            for (int k = 0; k < 3; k++) { pairs[p].lo[slot][k] = f_down(n.box.lo[k]); pairs[p].hi[slot][k] = f_up(n.box.hi[k]); }
            if (n.count) { pairs[p].child[slot] = leaf_first[ch[slot]]; pairs[p].meta[slot] = ((n.kind + 1u) << 16) | n.count; }
            else { pairs[p].child[slot] = pair_of[ch[slot]]; pairs[p].meta[slot] = 0; }
        }
    }
    void empty_child(uint32_t pair, int slot) {
        for (int k = 0; k < 3; k++) { pairs[pair].lo[slot][k] = 0.f; pairs[pair].hi[slot][k] = 0.f; }
        pairs[pair].child[slot] = 0;
        pairs[pair].meta[slot] = (1u << 16) | 0u;  // leaf with zero primitives
    }
    // ---- 4-wide nodes: collapse of the binary tree (largest-area internal child is opened first) ----
    static double area(const zr::BuildBox& b) {
        double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    // 8-bit planes of one axis: origin + q * scale, as a real number, must not exceed lo (lower plane) and must
    // reach hi (upper plane).  origin is a float, scale a power of two not smaller than 2^-30 |origin|, so the sum is
    // exact in long double and the comparison is the real one.
    static long double plane(float origin, float scale, long q) { return (long double)origin + (long double)q * (long double)scale; }
    static bool quant_axis(const double* lo, const double* hi, int n, float& origin, float& scale, uint8_t* qlo, uint8_t* qhi) {
        double mn = lo[0], mx = hi[0];
        for (int k = 1; k < n; k++) { mn = std::min(mn, lo[k]); mx = std::max(mx, hi[k]); }
        if (!std::isfinite(mn) || !std::isfinite(mx) || std::fabs(mn) > 1e30 || std::fabs(mx) > 1e30) return false;
        origin = f_down(mn);
        const double ext = mx - (double)origin;
        int e = ext > 0 ? (int)std::ceil(std::log2(ext / 255.0)) : -100;
        const int emin = origin != 0.0f ? std::max(-100, std::ilogb(origin) - 30) : -100;
        if (e < emin) e = emin;
        for (int tries = 0; tries < 64; tries++, e++) {
            scale = std::ldexp(1.0f, e);
            bool ok = true;
            for (int k = 0; k < n && ok; k++) {
                long ql = (long)std::floor((lo[k] - (double)origin) / (double)scale);
                ql = std::min(255l, std::max(0l, ql));
                while (ql > 0 && plane(origin, scale, ql) > (long double)lo[k]) ql--;
                if (plane(origin, scale, ql) > (long double)lo[k]) ok = false;
                long qh = (long)std::ceil((hi[k] - (double)origin) / (double)scale);
                qh = std::min(255l, std::max(0l, qh));
                while (qh < 255 && plane(origin, scale, qh) < (long double)hi[k]) qh++;
                if (plane(origin, scale, qh) < (long double)hi[k]) ok = false;
                qlo[k] = (uint8_t)ql; qhi[k] = (uint8_t)qh;
            }
            if (ok) return true;
        }
        return false;
    }
    double open_ratio = 1.25;  // a child is not opened when that would put a box on the grid with more than this times its true area
    zr::NodeF root{};
    size_t n_kept_closed = 0;
    // quantises the boxes of `kids` into nq; false when a box cannot be represented; *worst = largest area inflation
    bool quantise(const int32_t* kids, int nk, zr::NodeQ& nq, double* worst) const {
        uint8_t ql[3][4] = {}, qh[3][4] = {};
        for (int k = 0; k < 6; k++) nq.q[k] = 0;
        for (int ax = 0; ax < 3; ax++) {
            double lo[4], hi[4];
            for (int k = 0; k < nk; k++) { lo[k] = br.nodes[kids[k]].box.lo[ax]; hi[k] = br.nodes[kids[k]].box.hi[ax]; }
            if (!quant_axis(lo, hi, nk, nq.origin[ax], nq.scale[ax], ql[ax], qh[ax])) return false;
            for (int k = 0; k < nk; k++) { nq.q[ax] |= (uint32_t)ql[ax][k] << (8 * k); nq.q[3 + ax] |= (uint32_t)qh[ax][k] << (8 * k); }
        }
        *worst = 1;
        for (int k = 0; k < nk; k++) {
            zr::BuildBox qb;
            for (int ax = 0; ax < 3; ax++) {
                qb.lo[ax] = (double)plane(nq.origin[ax], nq.scale[ax], ql[ax][k]);
                qb.hi[ax] = (double)plane(nq.origin[ax], nq.scale[ax], qh[ax][k]);
            }
            const double at = area(br.nodes[kids[k]].box), aq = area(qb);
            const double r = at > 0 ? aq / at : (aq > 0 ? 1e300 : 1.0);
            if (!(r <= *worst)) *worst = r;
        }
        return true;
    }
    std::atomic<bool> quant_ok_a{true};   // false: a box below the root is not finite (the caller falls back to variant 0)
    bool quant_ok = true;
    std::atomic<size_t> kept_closed_a{0};
    // ---- 4-wide nodes.  Which children a node takes depends on quantisation trials, so a node's shape is only known once it is
    // planned; planning is level-synchronous — every node of a level in parallel, the inner children forming the next level — and
    // a serial pre-order walk then numbers the nodes and fills in the child references. ----
    struct QuadPlan { int32_t kids[4]; int nk; zr::NodeQ nq; };
    zr::RawArray<QuadPlan> plan;         // per build node (only quad roots are filled; not zero-filled)
    void plan_quad(int32_t node_id, bool is_root, QuadPlan& qp) const {
        int32_t* kids = qp.kids; int nk = 0;
        zr::NodeQ nq{};
        for (int ax = 0; ax < 3; ax++) nq.scale[ax] = 1;
        if (br.nodes[node_id].count) kids[nk++] = node_id;  // a world that is a single leaf
        else {
            nk = 2;
            kids[0] = br.nodes[node_id].left; kids[1] = br.nodes[node_id].right;
            double worst = 1;
            if (!is_root && !quantise(kids, nk, nq, &worst)) const_cast<Flattener*>(this)->quant_ok_a = false;
            while (nk < 4) {
                // open the inner child with the largest area, unless the grid of the wider node would be too coarse
                // for one of the boxes (then the child keeps its own node, whose grid fits its own children)
                int best = -1; double ba = -1;
                for (int k = 0; k < nk; k++) if (br.nodes[kids[k]].count == 0 && area(br.nodes[kids[k]].box) > ba) { ba = area(br.nodes[kids[k]].box); best = k; }
                if (best < 0) break;
                int32_t trial[4];
                for (int k = 0; k < nk; k++) trial[k] = kids[k];
                trial[best] = br.nodes[kids[best]].left; trial[nk] = br.nodes[kids[best]].right;
                if (!is_root) {
                    zr::NodeQ tq = nq; double w = 1;
                    if (!quantise(trial, nk + 1, tq, &w)) { const_cast<Flattener*>(this)->quant_ok_a = false; break; }
                    if (w > open_ratio && w > worst) { const_cast<Flattener*>(this)->kept_closed_a++; break; }
                    nq = tq; worst = w;
                }
                for (int k = 0; k <= nk; k++) kids[k] = trial[k];
                nk++;
            }
        }
        qp.nk = nk; qp.nq = nq;
    }
    // Numbering: pre-order, as a serial depth-first emit would number them — node index = parent's index + 1 + the sizes of the
    // subtrees of its earlier inner siblings (the root has none: its first child is node 0).  Subtree sizes come from a pass over the
    // planned levels bottom-up, indices from a pass top-down, the records are written by all threads.
    std::vector<std::vector<int32_t>> levels;     // planned quad roots, level by level (levels[0] = {root})
    zr::RawArray<uint32_t> q_size, q_index;       // per build node: quads in its subtree (itself included), its own index
    void plan_quads(int32_t root_id) {
        plan.allocate(br.nodes.size());
        levels.clear();
        levels.push_back(std::vector<int32_t>{root_id});
        bool first = !root_in_array;
        for (;;) {
            const std::vector<int32_t>& level = levels.back();
            const int T = std::max(1, threads);
            std::vector<std::vector<int32_t>> out((size_t)T);
            std::atomic<int> slot{0};
            parallel_for(level.size(), 256, [&](size_t a2, size_t b2) {
                std::vector<int32_t>& mine = out[(size_t)slot.fetch_add(1)];
                for (size_t i = a2; i < b2; i++) {
                    QuadPlan& qp = plan[level[i]];
                    plan_quad(level[i], first, qp);
                    for (int k = 0; k < qp.nk; k++) if (br.nodes[qp.kids[k]].count == 0) mine.push_back(qp.kids[k]);
                }
            });
            std::vector<int32_t> next;
            for (auto& v : out) next.insert(next.end(), v.begin(), v.end());
            first = false;
            if (next.empty()) break;
            levels.push_back(std::move(next));
        }
    }
    void number_quads(int32_t root_id) {
        q_size.allocate(br.nodes.size()); q_index.allocate(br.nodes.size());
        for (size_t l = levels.size(); l-- > 0;) {   // bottom-up: subtree sizes
            const std::vector<int32_t>& level = levels[l];
            parallel_for(level.size(), 2048, [&](size_t a2, size_t b2) {
                for (size_t i = a2; i < b2; i++) {
                    const QuadPlan& qp = plan[level[i]];
                    uint32_t n = 1;
                    for (int k = 0; k < qp.nk; k++) if (br.nodes[qp.kids[k]].count == 0) n += q_size[qp.kids[k]];
                    q_size[level[i]] = n;
                }
            });
        }
        const size_t n_quads = (size_t)q_size[root_id] - (root_in_array ? 0 : 1);   // the world's root travels in the kernel arguments
        q_index[root_id] = root_in_array ? 0u : 0xFFFFFFFFu;                        // ... so that its first child becomes node 0
        quads.resize(n_quads);
        for (size_t l = 0; l < levels.size(); l++) {           // top-down: indices, and the records themselves
            const std::vector<int32_t>& level = levels[l];
            parallel_for(level.size(), 1024, [&](size_t a2, size_t b2) {
                for (size_t i = a2; i < b2; i++) {
                    const int32_t node_id = level[i];
                    const QuadPlan& qp = plan[node_id];
                    uint32_t refs[4] = {ZR_REF_EMPTY, ZR_REF_EMPTY, ZR_REF_EMPTY, ZR_REF_EMPTY};
                    uint32_t next = q_index[node_id] + 1u;
                    for (int k = 0; k < qp.nk; k++) {
                        const zr::BuildNode& n = br.nodes[qp.kids[k]];
                        if (n.count) refs[k] = ZR_REF_LEAF | ((uint32_t)n.kind << 28) | ((uint32_t)(n.count - 1u) << 24) | leaf_first[qp.kids[k]];
                        else { refs[k] = next; q_index[qp.kids[k]] = next; next += q_size[qp.kids[k]]; }
                    }
                    if (l == 0 && !root_in_array) {
                        for (int k = 0; k < qp.nk; k++) {
                            const zr::BuildBox& bb = br.nodes[qp.kids[k]].box;
                            root.lox[k] = f_down(bb.lo[0]); root.loy[k] = f_down(bb.lo[1]); root.loz[k] = f_down(bb.lo[2]);
                            root.hix[k] = f_up(bb.hi[0]); root.hiy[k] = f_up(bb.hi[1]); root.hiz[k] = f_up(bb.hi[2]);
                        }
                        for (int k = 0; k < 4; k++) root.ref[k] = refs[k];
                    } else {
                        zr::NodeQ nq = qp.nq;
                        for (int k = 0; k < 4; k++) nq.ref[k] = refs[k];
                        quads[q_index[node_id]] = nq;
                    }
                }
            });
        }
        quad_depth = (int)levels.size() - 1;
        quant_ok = quant_ok_a.load(); n_kept_closed = kept_closed_a.load();
    }
    // Worst-case number of entries the EXTEND kernel's per-lane stack holds for this 4-wide tree: visiting a node whose
    // nk children are all hit pushes nk - 1 of them and descends into the nearest (any child can be the nearest), or
    // pushes all nk when the nearest is a leaf and the lane already holds a postponed leaf (zr_stream.hip).
    // demand(node) = max(nk, max over inner children c of nk - 1 + demand(c)); exact, by DFS over the emitted nodes.
    uint32_t demand_of(const uint32_t refs[4]) const {
        uint32_t nk = 0, best = 0;
        for (int k = 0; k < 4; k++) if (refs[k] != ZR_REF_EMPTY) nk++;
        for (int k = 0; k < 4; k++) {
            if (refs[k] == ZR_REF_EMPTY) continue;
            if (refs[k] & ZR_REF_LEAF) {   // a placed run: one sentinel entry, then the run's own tree on the same stack (zr_stream.hip, level 3)
                if (((refs[k] >> 28) & 7u) == ZR_KIND_INSTANCE && !run_demand.empty() && !insts.empty()) {
                    const uint32_t g = inst_group[refs[k] & 0xFFFFFFu];
                    best = std::max(best, 1u + run_demand[g]);
                }
                continue;
            }
            best = std::max(best, demand_of(quads[refs[k]].ref));
        }
        return std::max(nk, nk ? nk - 1 + best : 0u);
    }
    uint32_t stack_demand() const { return demand_of(root.ref); }
    void run() {
        const bool stats = std::getenv("ZR_COMMIT_STATS") != nullptr;
        auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_ph = now_s();
        auto ph = [&](const char* what) { if (stats) { const double t = now_s(); std::fprintf(stderr, "[zr] flatten: %-18s %.1f ms\n", what, (t - t_ph) * 1e3); t_ph = t; } };
        {
            unsigned hw = std::thread::hardware_concurrency();
            if (const char* e = std::getenv("ZR_BVH_THREADS")) hw = (unsigned)std::max(1, std::atoi(e));
            threads = (int)std::max(1u, std::min(32u, hw));
        }
        if (threads > 1 && br.nodes.size() > 65536) pool.reset(new Pool(threads - 1));
        struct Unpool { std::unique_ptr<Pool>& p; ~Unpool() { p.reset(); } } unpool{pool};   // the workers end with run()
        leaf_first.allocate(br.nodes.size());   // fresh pages: zero
        if (baked) for (uint8_t b : *baked) if (b) n_baked++;
        if (br.nodes.empty()) {
            pairs.allocate(1); empty_child(0, 0); empty_child(0, 1);
            for (int k = 0; k < 4; k++) root.ref[k] = ZR_REF_EMPTY;
            return;
        }
        // 1. indices
        pair_of.allocate(br.nodes.size());
        if (br.nodes[0].count) { leaf_first[0] = 0; cnt[br.nodes[0].kind & 7] = br.nodes[0].count; leaves.push_back(0); }   // the whole world fits one leaf
        else index_nodes();
        ph("index pass");
        // 2. array sizes: the leaf ranges, then the primitives inside media and wrapper chains
        size_t x_sph = 0, x_tri = 0, x_cube = 0, x_media = 0;
        auto count_inner = [&](uint32_t type, uint32_t idx, auto&& self) -> void {
            if (type == ZR_PRIM_SPHERE) x_sph++; else if (type == ZR_PRIM_TRIANGLE) x_tri++; else if (type == ZR_PRIM_CUBE) x_cube++;
            else { x_media++; self(s.media[idx].boundary_type, s.media[idx].boundary_index, self); }
        };
        const bool any_compound = cnt[ZR_PRIM_MEDIUM] + cnt[ZR_KIND_WRAPPED] != 0;   // (a million-leaf scan for nothing otherwise)
        if (any_compound) for (int32_t lf : leaves) {
            const zr::BuildNode& n = br.nodes[lf];
            if (n.kind != ZR_PRIM_MEDIUM && n.kind != ZR_KIND_WRAPPED) continue;
            for (uint32_t k = 0; k < n.count; k++) {
                const zr_object& o = objs[br.order[n.first + k]];
                if (n.kind == ZR_PRIM_MEDIUM) count_inner(s.media[o.index].boundary_type, s.media[o.index].boundary_index, count_inner);
                else count_inner(o.type, o.index, count_inner);
            }
        }
        n_sph = cnt[ZR_PRIM_SPHERE]; n_tri = cnt[ZR_PRIM_TRIANGLE]; n_cube = cnt[ZR_PRIM_CUBE]; n_pcube = cnt[ZR_KIND_PCUBE];
        n_media = cnt[ZR_PRIM_MEDIUM]; n_wrapped = cnt[ZR_KIND_WRAPPED];
        size_t run_pair_total = 0;
        if (runs) for (size_t g = 0; g < runs->size(); g++) { x_tri += s.groups[g].triangle_count; run_pair_total += run_pairs((*runs)[g]); }
        insts.allocate(cnt[ZR_KIND_INSTANCE]); inst_group.assign(cnt[ZR_KIND_INSTANCE], 0);
        const size_t main_pairs = std::max<size_t>(1, inner.size());
        pairs.allocate(main_pairs + run_pair_total);
        if (runs) {   // where each group's subtree will start (the records follow the world's own)
            run_root.resize(runs->size()); run_tri_base.assign(runs->size(), 0);
            size_t at = main_pairs;
            for (size_t g = 0; g < runs->size(); g++) { run_root[g] = (uint32_t)at; at += run_pairs((*runs)[g]); }
        }
        spheres.allocate((n_sph + x_sph) * 4); sphere_mat.allocate(n_sph + x_sph);
        tri_v.allocate((n_tri + x_tri) * ZR_TRI_STRIDE); tri_s.allocate((n_tri + x_tri) * 20);
        cubes.allocate((n_cube + x_cube) * 6); cube_mat.allocate(n_cube + x_cube);
        pcubes.allocate(n_pcube * ZR_PCUBE_STRIDE); pcube_mat.allocate(n_pcube);
        media.allocate(n_media + x_media); wrapped.allocate(n_wrapped);
        // 3. leaf primitives of the plain kinds: all threads
        parallel_for(leaves.size(), 2048, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                const zr::BuildNode& n = br.nodes[leaves[i]];
                if (n.kind == ZR_PRIM_MEDIUM || n.kind == ZR_KIND_WRAPPED) continue;
                for (uint32_t k = 0; k < n.count; k++) put_leaf_object(br.order[n.first + k], (size_t)leaf_first[leaves[i]] + k);
            }
        });
        if (runs) for (size_t g = 0; g < runs->size(); g++) emit_run((uint32_t)g, run_root[g]);   // serial: runs are shared, hence few
        ph("primitive records");
        // media and wrapped objects, with what they contain: serial, in emit order
        if (any_compound) for (int32_t lf : leaves) {
            const zr::BuildNode& n = br.nodes[lf];
            if (n.kind != ZR_PRIM_MEDIUM && n.kind != ZR_KIND_WRAPPED) continue;
            for (uint32_t k = 0; k < n.count; k++) {
                const zr_object& o = objs[br.order[n.first + k]];
                const size_t di = (size_t)leaf_first[lf] + k;
                if (n.kind == ZR_PRIM_MEDIUM) put_medium(di, o.index);
                else {
                    zr::DWrapped w{};
                    w.type = o.type; w.chain_first = o.chain_first; w.chain_count = o.chain_count;
                    w.index = append_inner(o.type, o.index);
                    wrapped[di] = w;
                }
            }
        }
        ph("media / wrapped");
        if (after_primitives) after_primitives();   // the primitive arrays are final: their upload can run beside the rest
        // 4. pair records: all threads
        if (inner.empty()) fill_leaf_root();
        else {
            parallel_for(inner.size(), 4096, [&](size_t a, size_t b) { for (size_t p = a; p < b; p++) fill_pair((uint32_t)p, inner[p]); });
        }
        ph("pair records");
        // 5. 4-wide nodes
        plan_quads(0);
        ph("4-wide plan");
        number_quads(0);
        ph("4-wide numbering");
        if (runs) { emit_run_quads(); ph("groups' 4-wide nodes"); }
    }
    // The 4-wide quantised nodes of every group, behind the world's own: planned and numbered by a flattener of the group's tree
    // (same collapse, same quantisation; its root is a stored node, not kernel arguments), then copied with the indices moved —
    // inner references by the group's first node, triangle references by the group's first triangle.  Placements get the root.
    void emit_run_quads() {
        run_qroot.assign(runs->size(), 0); run_demand.assign(runs->size(), 0);
        const std::vector<zr_object> none;
        for (size_t g = 0; g < runs->size(); g++) {
            const zr::BuildResult& rb = (*runs)[g];
            Flattener sub{s, none, rb};
            sub.threads = 1; sub.open_ratio = open_ratio; sub.root_in_array = true;
            sub.leaf_first.allocate(rb.nodes.size()); sub.pair_of.allocate(rb.nodes.size());
            const uint32_t base = (uint32_t)quads.size();
            run_qroot[g] = base;
            if (rb.nodes[0].count) {   // the whole run is one leaf: a node with one child
                zr::NodeQ nq{};
                int32_t kid = 0; double w = 1;
                for (int ax = 0; ax < 3; ax++) nq.scale[ax] = 1;
                if (!sub.quantise(&kid, 1, nq, &w)) quant_ok_a = false;
                for (int k = 0; k < 4; k++) nq.ref[k] = ZR_REF_EMPTY;
                nq.ref[0] = ZR_REF_LEAF | ((uint32_t)ZR_PRIM_TRIANGLE << 28) | ((uint32_t)(rb.nodes[0].count - 1u) << 24) | run_tri_base[g];
                quads.push_back(nq);
                run_demand[g] = 1;
                continue;
            }
            sub.index_nodes();
            sub.plan_quads(0);
            sub.number_quads(0);
            if (!sub.quant_ok) quant_ok_a = false;
            for (zr::NodeQ nq : sub.quads) {
                for (int k = 0; k < 4; k++) {
                    if (nq.ref[k] == ZR_REF_EMPTY) continue;
                    if (nq.ref[k] & ZR_REF_LEAF) nq.ref[k] += run_tri_base[g];   // (the low 24 bits: the first primitive)
                    else nq.ref[k] += base;
                }
                quads.push_back(nq);
            }
            run_demand[g] = sub.demand_of(sub.quads[0].ref) + 0u;
        }
        quant_ok = quant_ok && quant_ok_a.load();
        for (size_t i = 0; i < insts.size(); i++) insts[i].pad_ = run_qroot[inst_group[i]];   // DInstance::qroot
    }
    void fill_leaf_root() {   // the whole world in one leaf: a pair whose second child is empty
        const zr::BuildNode& n = br.nodes[0];
        for (int k = 0; k < 3; k++) { pairs[0].lo[0][k] = f_down(n.box.lo[k]); pairs[0].hi[0][k] = f_up(n.box.hi[k]); }
        pairs[0].child[0] = leaf_first[0]; pairs[0].meta[0] = ((n.kind + 1u) << 16) | n.count;
        empty_child(0, 1);
    }
};

double env_double(const char* name, double dflt) {
    const char* v = std::getenv(name);
    return v && *v ? std::atof(v) : dflt;
}

struct CommitSummary {   // what the shared end of a commit needs to know about the tree either builder produced
    zr::NodeF root{}; bool quant_ok = true;
    size_t n_pairs = 0, n_quads = 0, n_sph = 0, n_tri = 0, n_cube = 0, n_pcube = 0, n_media = 0, n_wrapped = 0, n_insts = 0, kept_closed = 0;
    bool plain_media = true;
    uint32_t stack_demand = 0; int quad_depth = 0, max_depth = 0, max_leaf = 4;
    uint32_t leaf_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // leaf objects per kind
    const char* builder = "";
};
// the tables every scene has (ops, materials, textures), the DScene the kernels receive, the EXTEND build the world needs
int finish_commit(zr_scene* s, const CommitSummary& cs, size_t n_objs) {
    int rc;
    if (std::getenv("ZR_QUANT_STATS")) std::fprintf(stderr, "[zr] 4-wide nodes: %zu quantised (64 B) + FP32 root; %zu children kept closed for the grid\n", cs.n_quads, cs.kept_closed);
    s->quad_ok = cs.quant_ok && cs.n_quads < (1u << 31) && cs.max_leaf <= 16 && cs.n_sph < (1u << 24) && cs.n_tri < (1u << 24) && cs.n_cube < (1u << 24) &&
                 cs.n_media < (1u << 24) && cs.n_wrapped < (1u << 24) && cs.n_pcube < (1u << 24) && cs.n_insts < (1u << 24);
    if ((rc = s->d_ops.upload(s->ops.data(), s->ops.size()))) return rc;
    {
        // zr_material::pad_ on the device copy: the material reads u/v/tangent (image texture anywhere in its
        // texture tree, or a bump map) -> the kernels compute those hit-record fields only then
        std::vector<zr_material> mats = s->materials;
        auto tex_uses_uv = [&](uint32_t id) {
            std::vector<uint32_t> todo{id}; int guard = 0;
            while (!todo.empty() && guard++ < 4096) {
                uint32_t t = todo.back(); todo.pop_back();
                if (t >= s->textures.size()) continue;
                const zr_texture& tx = s->textures[t];
                if (tx.kind >= ZR_TEX_IMAGE_U8) return true;
                if (tx.kind == ZR_TEX_CHECKER) { todo.push_back(tx.odd); todo.push_back(tx.even); }
            }
            return guard >= 4096;
        };
        for (zr_material& m : mats) m.pad_ = (m.bump_tex != ZR_NO_TEXTURE || (m.kind != ZR_MAT_DIELECTRIC && tex_uses_uv(m.tex))) ? 1u : 0u;
        if ((rc = s->d_mats.upload(mats))) return rc;
    }
    if ((rc = s->d_texs.upload(s->textures))) return rc;
    if ((rc = s->d_texels.upload(s->texels.data(), s->texels.size()))) return rc;

    zr::DScene& d = s->ds;
    d.nodes = s->d_nodes.p; d.quads = s->d_quads.p;
    d.spheres = s->d_spheres.p; d.sphere_mat = s->d_sphere_mat.p;
    d.tri_v = s->d_tri_v.p; d.tri_s = s->d_tri_s.p;
    d.cubes = s->d_cubes.p; d.cube_mat = s->d_cube_mat.p;
    d.pcubes = s->d_pcubes.p; d.pcube_mat = s->d_pcube_mat.p;
    d.media = s->d_media.p; d.wrapped = s->d_wrapped.p; d.insts = s->d_insts.p; d.ops = s->d_ops.p;
    d.mats = s->d_mats.p; d.texs = s->d_texs.p; d.texels = s->d_texels.p;
    d.n_mats = (uint32_t)s->materials.size();
    d.mat_kinds = 0;
    for (const zr_material& m : s->materials) d.mat_kinds |= 1u << m.kind;
    d.root = cs.root;
    s->leaf_objects = 0;
    for (int k = 0; k < 8; k++) { d.leaf_cnt[k] = cs.leaf_cnt[k]; s->leaf_objects += cs.leaf_cnt[k]; }
    // a small world's objects for the fused kernel's arguments (zr_launch.h: FusedObjs): read back from the arrays just built,
    // whichever builder made them (a few hundred bytes)
    s->fused_ok = false;
    if (s->leaf_objects > 0 && s->leaf_objects <= ZR_FUSED_OBJECTS && cs.leaf_cnt[ZR_KIND_INSTANCE] == 0) {
        zr::FusedObjs fo{};
        auto take = [&](uint32_t kind, const double* d_src, size_t stride, size_t doubles) -> int {
            for (uint32_t i = 0; i < cs.leaf_cnt[kind]; i++) {
                fo.kind[fo.n] = kind; fo.index[fo.n] = i;
                HIP_OK(hipMemcpy(fo.rec[fo.n], d_src + (size_t)i * stride, doubles * sizeof(double), hipMemcpyDeviceToHost));
                fo.n++;
            }
            return ZR_OK;
        };
        if ((rc = take(ZR_PRIM_SPHERE, s->d_spheres.p, 4, 4)) || (rc = take(ZR_PRIM_TRIANGLE, s->d_tri_v.p, ZR_TRI_STRIDE, 9)) ||
            (rc = take(ZR_PRIM_CUBE, s->d_cubes.p, 6, 6)) || (rc = take(ZR_KIND_PCUBE, s->d_pcubes.p, ZR_PCUBE_STRIDE, ZR_PCUBE_STRIDE))) return rc;
        std::vector<zr::DMedium> hm(cs.leaf_cnt[ZR_PRIM_MEDIUM]);
        if (!hm.empty()) HIP_OK(hipMemcpy(hm.data(), s->d_media.p, hm.size() * sizeof(zr::DMedium), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < hm.size(); i++) {
            if (hm[i].chain_count != 0) continue;   // a wrapped boundary: tested through the scene's arrays (level 2)
            fo.kind[fo.n] = ZR_PRIM_MEDIUM; fo.index[fo.n] = i;
            const bool sph = hm[i].btype == ZR_PRIM_SPHERE;
            HIP_OK(hipMemcpy(fo.rec[fo.n], sph ? s->d_spheres.p + (size_t)hm[i].bindex * 4 : s->d_cubes.p + (size_t)hm[i].bindex * 6, (sph ? 4 : 6) * sizeof(double), hipMemcpyDeviceToHost));
            fo.rec[fo.n][6] = hm[i].neg_inv_density;
            const uint64_t idb = hm[i].id, tb = hm[i].btype;
            std::memcpy(&fo.rec[fo.n][7], &idb, 8); std::memcpy(&fo.rec[fo.n][8], &tb, 8);
            fo.n++;
        }
        bool scaled = false;   // the fused kernel's placed-cube code carries no scale (zr_device.h pcube_ray<false>): such a world takes the pipeline
        for (uint32_t i = 0; i < fo.n; i++) if (fo.kind[i] == ZR_KIND_PCUBE && fo.rec[i][15] != 0.0) scaled = true;
        s->fused = fo; s->fused_ok = !scaled;
    }
    {   // which build of the EXTEND kernel this world needs (zr_stream.hip)
        if (cs.n_insts) s->leaf_level = 3;   // placed runs of triangles: the build with the nested walk
        else if (cs.n_wrapped || !cs.plain_media) s->leaf_level = 2;
        else if (cs.n_cube || cs.n_pcube || cs.n_media) s->leaf_level = 1;
        else s->leaf_level = 0;
        const int force = (int)env_double("ZR_EXTEND_LEVEL", -1);
        if (force > s->leaf_level && force <= 3) s->leaf_level = force;
        // SHADE's lean build (zr_device.h: lean_rec / lean_shade): a world of bare triangles and spheres whose materials are lambertian / metal / dielectric / light
        // over solid-colour textures, no bump maps — nothing in it reads u, v, a tangent, an image or a wrapper chain
        bool lean = s->leaf_level == 0 && env_double("ZR_SHADE_LEAN", 1) != 0;
        for (const zr_material& m : s->materials) {
            if (m.kind != ZR_MAT_LAMBERTIAN && m.kind != ZR_MAT_METAL && m.kind != ZR_MAT_DIELECTRIC && m.kind != ZR_MAT_LIGHT) lean = false;
            if (m.bump_tex != ZR_NO_TEXTURE) lean = false;
            if (m.kind != ZR_MAT_DIELECTRIC && (m.tex >= s->textures.size() || s->textures[m.tex].kind != ZR_TEX_SOLID)) lean = false;
        }
        d.shade_lean = lean ? 1u : 0u;
    }
    s->stack_demand = cs.stack_demand;
    if (std::getenv("ZR_QUANT_STATS")) std::fprintf(stderr, "[zr] 4-wide tree: depth %d, worst-case traversal stack %u entries\n", cs.quad_depth, s->stack_demand);
    s->stats[0] = cs.n_pairs; s->stats[1] = (uint64_t)cs.max_depth; s->stats[2] = n_objs;
    s->stats[3] = cs.n_pairs * sizeof(zr::NodePair) + cs.n_quads * sizeof(zr::NodeQ) + (cs.n_sph * 4 + cs.n_tri * (ZR_TRI_STRIDE + 20) + cs.n_cube * 6 + cs.n_pcube * ZR_PCUBE_STRIDE) * 8 +
                  (cs.n_sph + cs.n_cube) * 4 + s->texels.size();
    s->builder = cs.builder;
    s->committed = true;
    if (s->borrowed) {   // the caller's arrays are not read again: forget them (a second commit needs a new zr_scene_set_*)
        s->spheres.drop(); s->sphere_mat.drop(); s->tri_v.drop(); s->tri_n.drop(); s->tri_mat.drop(); s->cubes.drop(); s->cube_mat.drop();
        s->media.drop(); s->ops.drop(); s->objects.drop(); s->texels.drop(); s->objects_set = false; s->borrowed = false; s->released = true;
    }
    return ZR_OK;
}


// ---- the commit with the tree built ON THE DEVICE (zr_build.hip) ------------------------------------------------------------------
// The scene's arrays go to the device as they are; boxes, Morton keys, sort, PLOC merging, leaf collapse, the 4-wide quantised
// nodes, the pair records and the primitive records in leaf order are all produced there.  The host classifies the world-list
// entries (a pass over 16-byte records), finishes the few compound objects (media, wrapped objects: each drags inner primitives
// behind the leaf ranges) and sizes the final arrays.  ZR_E_STATE from here means "this input is for the host builder" (a tree
// deeper than the traversal stack, coordinates beyond 1e18): the caller falls back.
constexpr int ZR_FALLBACK_HOST = 1;
int commit_device(zr_scene* s, const std::vector<zr_object>& objs, bool commit_stats, CommitSummary& cs) {
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_phase = now_s();
    auto phase = [&](const char* what) { if (commit_stats) { const double t = now_s(); std::fprintf(stderr, "[zr] commit(device): %-22s %.1f ms\n", what, (t - t_phase) * 1e3); t_phase = t; } };
    const uint32_t n = (uint32_t)objs.size();
    hipStream_t st = s->ctx->stream;
    int rc;
    // every group's tree is a build of its own (a few dozen launches and a handful of synchronisations, ~1 ms however small the run):
    // a world of very many small groups is the host builder's, which builds them on its threads
    if ((double)s->groups.size() > env_double("ZR_BVH_DEVICE_MAX_GROUPS", 256)) {
        std::fprintf(stderr, "[zr] device BVH build: %zu groups of triangles: host builder\n", s->groups.size());
        return ZR_FALLBACK_HOST;
    }
    // 1. classification + array sizes
    std::vector<uint8_t> code(n);
    const bool bake = env_double("ZR_BAKE_TRIANGLES", 1) != 0;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        unsigned hw = std::thread::hardware_concurrency();
        const int T = n < 65536 ? 1 : (int)std::max(1u, std::min(16u, hw));
        std::vector<std::array<uint32_t, 8>> part((size_t)T, std::array<uint32_t, 8>{});
        auto work = [&](int t, size_t k0, size_t k1) {
            std::array<uint32_t, 8> c{};
            for (size_t k = k0; k < k1; k++) { uint32_t kind; uint8_t bk; classify_object(*s, objs[k], bake, kind, bk); code[k] = (uint8_t)(kind | (bk << 4)); c[kind & 7]++; }
            part[(size_t)t] = c;
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t, (size_t)n * t / T, (size_t)n * (t + 1) / T);
        work(0, 0, (size_t)n / T);
        for (auto& x : th) x.join();
        for (auto& c : part) for (int k = 0; k < 8; k++) cnt[k] += c[k];
    }
    size_t x_sph = 0, x_tri = 0, x_cube = 0, x_media = 0;   // primitives inside media and wrapper chains: behind the leaf ranges
    auto count_inner = [&](uint32_t type, uint32_t idx, auto&& self) -> void {
        if (type == ZR_PRIM_SPHERE) x_sph++; else if (type == ZR_PRIM_TRIANGLE) x_tri++; else if (type == ZR_PRIM_CUBE) x_cube++;
        else { x_media++; self(s->media[idx].boundary_type, s->media[idx].boundary_index, self); }
    };
    if (cnt[ZR_PRIM_MEDIUM] + cnt[ZR_KIND_WRAPPED])
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t kind = code[k] & 7u;
            if (kind == ZR_PRIM_MEDIUM) count_inner(s->media[objs[k].index].boundary_type, s->media[objs[k].index].boundary_index, count_inner);
            else if (kind == ZR_KIND_WRAPPED) count_inner(objs[k].type, objs[k].index, count_inner);
        }
    size_t group_tris = 0;
    for (const zr_group& g : s->groups) group_tris += g.triangle_count;
    const size_t n_sph = cnt[ZR_PRIM_SPHERE] + x_sph, n_tri = cnt[ZR_PRIM_TRIANGLE] + group_tris + x_tri, n_cube = cnt[ZR_PRIM_CUBE] + x_cube;
    const size_t n_pcube = cnt[ZR_KIND_PCUBE], n_media = cnt[ZR_PRIM_MEDIUM] + x_media, n_wrapped = cnt[ZR_KIND_WRAPPED], n_insts = cnt[ZR_KIND_INSTANCE];
    phase("classify");
    // 2. the scene as given -> device (freed with this call), the final primitive arrays allocated.  The large arrays are pinned for the
    // copy (hipHostRegister: 2.4 ms per 160 MB on the GPU box, then 57 GB/s instead of the ~10 GB/s of a first pageable copy,
    // profiles/r3_affine_ab.txt) and travel asynchronously on the build's stream, under the classification above... and below
    DevBuf<double> r_sph, r_tri_v, r_tri_n, r_cubes, r_gbox;
    DevBuf<uint32_t> r_sph_mat, r_tri_mat, r_cube_mat, d_inst_group, d_run_demand, d_run_root, d_run_qroot;
    DevBuf<zr_medium> r_media; DevBuf<zr_object> r_objs; DevBuf<uint8_t> r_code;
    // (an early return leaves copies in flight: they are waited for before their source pages are unpinned)
    struct Pinned { hipStream_t st; std::vector<void*> p; ~Pinned() { if (!p.empty()) (void)hipStreamSynchronize(st); for (void* q : p) (void)hipHostUnregister(q); } } pinned{st, {}};
    auto send = [&](auto& buf, const auto* src, size_t count) -> int {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(src)>>;
        int r = buf.alloc(count);
        if (r || count == 0) return r;
        const size_t bytes = count * sizeof(T);
        if (bytes >= (4u << 20) && hipHostRegister((void*)src, bytes, hipHostRegisterDefault) == hipSuccess) {
            pinned.p.push_back((void*)src);
            HIP_OK(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));
        } else {
            (void)hipGetLastError();
            HIP_OK(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice));
        }
        return ZR_OK;
    };
    if ((rc = send(r_tri_v, s->tri_v.data(), s->tri_v.size())) || (rc = send(r_tri_n, s->tri_n.data(), s->tri_n.size())) ||
        (rc = send(r_objs, objs.data(), objs.size())) || (rc = send(r_tri_mat, s->tri_mat.data(), s->tri_mat.size())) ||
        (rc = send(r_sph, s->spheres.data(), s->spheres.size())) || (rc = send(r_sph_mat, s->sphere_mat.data(), s->sphere_mat.size())) ||
        (rc = send(r_cubes, s->cubes.data(), s->cubes.size())) || (rc = send(r_cube_mat, s->cube_mat.data(), s->cube_mat.size())) ||
        (rc = send(r_media, s->media.data(), s->media.size())) || (rc = s->d_ops.upload(s->ops.data(), s->ops.size())) || (rc = send(r_code, code.data(), code.size()))) return rc;
    if ((rc = s->d_spheres.alloc(n_sph * 4)) || (rc = s->d_sphere_mat.alloc(n_sph)) || (rc = s->d_tri_v.alloc(n_tri * ZR_TRI_STRIDE)) || (rc = s->d_tri_s.alloc(n_tri * 20)) ||
        (rc = s->d_cubes.alloc(n_cube * 6)) || (rc = s->d_cube_mat.alloc(n_cube)) || (rc = s->d_pcubes.alloc(n_pcube * ZR_PCUBE_STRIDE)) || (rc = s->d_pcube_mat.alloc(n_pcube)) ||
        (rc = s->d_insts.alloc(n_insts)) || (rc = d_inst_group.alloc(n_insts))) return rc;
    phase("upload as given");
    zr::BuildSceneIn in;
    in.spheres = r_sph.p; in.sphere_mat = r_sph_mat.p; in.tri_v = r_tri_v.p; in.tri_n = r_tri_n.p; in.tri_mat = r_tri_mat.p;
    in.cubes = r_cubes.p; in.cube_mat = r_cube_mat.p; in.media = r_media.p; in.ops = s->d_ops.p;
    zr::BuildParams bp;
    bp.ct = (float)env_double("ZR_BVH_COST_TRAVERSE", 1.0);
    const double ck[8] = {env_double("ZR_BVH_COST_SPHERE", 1.0), env_double("ZR_BVH_COST_TRI", 1.5), env_double("ZR_BVH_COST_CUBE", 1.0),
                          env_double("ZR_BVH_COST_MEDIUM", 3.0), env_double("ZR_BVH_COST_WRAPPED", 3.0), env_double("ZR_BVH_COST_PCUBE", 1.5),
                          env_double("ZR_BVH_COST_GROUP", 16.0), 1};
    for (int k = 0; k < 8; k++) bp.ck[k] = (float)ck[k];
    bp.max_leaf = (int)env_double("ZR_BVH_MAX_LEAF", 4);
    const int big = (int)env_double("ZR_BVH_MAX_LEAF_BIG", 1);
    const int leaf_cap[8] = {0, 0, big, big, big, big, 1, 0};
    for (int k = 0; k < 8; k++) bp.leaf_cap[k] = leaf_cap[k];
    bp.open_ratio = (float)env_double("ZR_BVH_OPEN_RATIO", 1.25);
    bp.radius = (int)env_double("ZR_BVH_PLOC_RADIUS", 16);
    // PLOC stops at n / 64 clusters (4096 ... 65536) and the host's SAH builder arranges those: the larger the SAH-built top, the closer
    // the walk comes to the host tree's, and the longer the host's pass takes (cfg3 EXTEND per frame against the host tree's: no top
    // +8.3 %, 16384 clusters +2.6 %, 65536 +2.3 %, for 17 / 19 / 28 ms of commit — the reference commits once per frame, so the default
    // is the 16384 a million objects get; profiles/r3_builders.txt).  ZR_BVH_TOP overrides (0: PLOC to the root)
    {
        const double top_env = env_double("ZR_BVH_TOP", -1);
        bp.top_clusters = top_env >= 0 ? (int)top_env : (int)std::min<size_t>(65536, std::max<size_t>(4096, (size_t)n / 64));
    }
    zr::BuildPrimOut out;
    out.spheres = s->d_spheres.p; out.sphere_mat = s->d_sphere_mat.p; out.tri_v = s->d_tri_v.p; out.tri_s = s->d_tri_s.p;
    out.cubes = s->d_cubes.p; out.cube_mat = s->d_cube_mat.p; out.pcubes = s->d_pcubes.p; out.pcube_mat = s->d_pcube_mat.p;
    out.insts = s->d_insts.p; out.inst_group = d_inst_group.p;
    auto builder = std::make_shared<zr::DeviceBuilder>(st);
    auto build_fail = [&](hipError_t e) {
        if (e == hipErrorInvalidValue) { std::fprintf(stderr, "[zr] device BVH build: %s\n", builder->error()); return (int)ZR_FALLBACK_HOST; }
        return fail(ZR_E_DEVICE, "device BVH build failed: %s (%s)", hipGetErrorString(e), builder->error());
    };
    // 3. the groups' trees (two-level BVH: one tree per shared run of triangles, in its own space)
    const size_t ng = s->groups.size();
    std::vector<zr::BuiltTree> runs(ng);
    std::vector<double> gbox(ng * 6);
    std::vector<uint32_t> run_demand(ng), run_tri_base(ng);
    {
        size_t at = cnt[ZR_PRIM_TRIANGLE];
        for (size_t g = 0; g < ng; g++) {
            const zr_group& grp = s->groups[g];
            run_tri_base[g] = (uint32_t)at; at += grp.triangle_count;
            zr::BuildPrimOut go = out;
            go.base[ZR_PRIM_TRIANGLE] = run_tri_base[g];
            hipError_t e = builder->build(in, nullptr, nullptr, grp.first_triangle, grp.triangle_count, bp, true, go, nullptr, ZR_STACK_DEPTH - 2, false, runs[g]);
            if (e != hipSuccess) return build_fail(e);
            for (int k = 0; k < 6; k++) gbox[g * 6 + k] = runs[g].box[k];
            run_demand[g] = runs[g].demand;
        }
    }
    if (ng) { if ((rc = r_gbox.upload(gbox)) || (rc = d_run_demand.upload(run_demand))) return rc; in.group_box = r_gbox.p; }
    phase("groups' trees");
    // 4. the world's tree
    zr::BuiltTree world;
    world.want_boxes = std::getenv("ZR_BUILD_CHECK") != nullptr;
    {
        hipError_t e = builder->build(in, r_objs.p, r_code.p, 0, n, bp, false, out, ng ? d_run_demand.p : nullptr, ZR_STACK_DEPTH - 2, commit_stats, world);
        if (e != hipSuccess) return build_fail(e);
    }
    if (commit_stats)
        std::fprintf(stderr, "[zr] device build: boxes+keys %.2f, sort %.2f, PLOC %.2f (%u iterations), order %.2f, 4-wide %.2f, pairs %.2f, emit %.2f ms; depth %u, %u pairs, %u quads\n",
                     world.ms[0], world.ms[1], world.ms[2], world.ploc_iterations, world.ms[3], world.ms[4], world.ms[5], world.ms[6], world.depth, world.n_pairs, world.n_quads);
    if (world.want_boxes) {   // self-check: every object's device box must contain the box the host's Boxer computes for it
        std::vector<zr::BuildBox> gb(ng);
        for (size_t g = 0; g < ng; g++) for (int k = 0; k < 3; k++) { gb[g].lo[k] = gbox[g * 6 + k]; gb[g].hi[k] = gbox[g * 6 + 3 + k]; }
        Boxer boxer{*s, &gb};
        size_t bad = 0;
        for (uint32_t k = 0; k < n; k++) {
            const zr::BuildBox hb = boxer.chain(objs[k].type, objs[k].index, objs[k].chain_first, objs[k].chain_count);
            const float* d = &world.dbg_boxes[(size_t)k * 8];
            bool ok = true;
            for (int a = 0; a < 3; a++) if (!((double)d[a] <= hb.lo[a]) || !((double)d[4 + a] >= hb.hi[a])) ok = false;
            if (!ok && bad++ < 8)
                std::fprintf(stderr, "[zr] BUILD_CHECK: object %u (type %u, chain %u): device box [%g %g %g | %g %g %g] does not contain the host's [%g %g %g | %g %g %g]\n", k, objs[k].type,
                             objs[k].chain_count, d[0], d[1], d[2], d[4], d[5], d[6], hb.lo[0], hb.lo[1], hb.lo[2], hb.hi[0], hb.hi[1], hb.hi[2]);
        }
        if (bad) return fail(ZR_E_DEVICE, "ZR_BUILD_CHECK: %zu of %u object boxes computed on the device do not contain the host's", bad, n);
    }
    for (int k = 0; k < 8; k++)
        if (world.cnt[k] != cnt[k]) return fail(ZR_E_DEVICE, "device BVH build: %u leaf primitives of kind %d, expected %u (internal error)", world.cnt[k], k, cnt[k]);
    phase("world tree");
    // 5. the scene's node arrays at their exact sizes: the world's records first, then every group's
    size_t n_pairs = world.n_pairs, n_quads = world.n_quads;
    std::vector<uint32_t> run_root(ng), run_qroot(ng);
    for (size_t g = 0; g < ng; g++) { run_root[g] = (uint32_t)n_pairs; run_qroot[g] = (uint32_t)n_quads; n_pairs += runs[g].n_pairs; n_quads += runs[g].n_quads; }
    if ((rc = s->d_nodes.alloc(n_pairs)) || (rc = s->d_quads.alloc(n_quads))) return rc;
    {
        hipError_t e = builder->relocate(world, s->d_nodes.p, 0, s->d_quads.p, 0);
        for (size_t g = 0; g < ng && e == hipSuccess; g++) e = builder->relocate(runs[g], s->d_nodes.p, run_root[g], s->d_quads.p, run_qroot[g]);
        if (e == hipSuccess && n_insts) {
            if ((rc = d_run_root.upload(run_root)) || (rc = d_run_qroot.upload(run_qroot))) return rc;
            e = builder->patch_instances(s->d_insts.p, d_inst_group.p, (uint32_t)n_insts, d_run_root.p, d_run_qroot.p);
        }
        if (e != hipSuccess) return fail(ZR_E_DEVICE, "device BVH build: %s", hipGetErrorString(e));
    }
    // 6. compound objects on the host: a medium's boundary, the object inside a wrapper chain (Flattener's own routines, on arrays
    // whose untouched pages cost nothing; only what they wrote is uploaded)
    bool plain_media = true;
    if ((rc = s->d_media.alloc(n_media)) || (rc = s->d_wrapped.alloc(n_wrapped))) return rc;
    if (n_media + n_wrapped) {
        static const zr::BuildResult no_tree;
        Flattener fl{*s, objs, no_tree};
        fl.spheres.allocate(n_sph * 4); fl.sphere_mat.allocate(n_sph);
        fl.tri_v.allocate(n_tri * ZR_TRI_STRIDE); fl.tri_s.allocate(n_tri * 20);
        fl.cubes.allocate(n_cube * 6); fl.cube_mat.allocate(n_cube);
        fl.media.allocate(n_media); fl.wrapped.allocate(n_wrapped);
        fl.n_sph = cnt[ZR_PRIM_SPHERE]; fl.n_tri = cnt[ZR_PRIM_TRIANGLE] + group_tris; fl.n_cube = cnt[ZR_PRIM_CUBE]; fl.n_media = cnt[ZR_PRIM_MEDIUM];
        const size_t b_sph = fl.n_sph, b_tri = fl.n_tri, b_cube = fl.n_cube;
        std::vector<std::pair<uint32_t, uint32_t>> todo;   // (index in its kind's array, object): leaf order, media before wrapped objects
        for (int pass = 0; pass < 2; pass++) {
            todo.clear();
            for (size_t k = 0; k + 1 < world.compound.size(); k += 2) {
                const uint32_t oi = world.compound[k], di = world.compound[k + 1];
                if (((code[oi] & 7u) == ZR_PRIM_MEDIUM) == (pass == 0)) todo.emplace_back(di, oi);
            }
            std::sort(todo.begin(), todo.end());
            for (const auto& [di, oi] : todo) {
                const zr_object& o = objs[oi];
                if (pass == 0) fl.put_medium(di, o.index);
                else {
                    zr::DWrapped w{};
                    w.type = o.type; w.chain_first = o.chain_first; w.chain_count = o.chain_count;
                    w.index = fl.append_inner(o.type, o.index);
                    fl.wrapped[di] = w;
                }
            }
        }
        if (fl.n_sph != n_sph || fl.n_tri != n_tri || fl.n_cube != n_cube || fl.n_media != n_media)
            return fail(ZR_E_DEVICE, "device BVH build: compound objects do not add up (internal error)");
        for (size_t k = 0; k < n_media; k++) if (fl.media[k].chain_count != 0) plain_media = false;
        auto up = [&](void* dst, const void* src, size_t bytes) -> int { if (bytes) HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); return ZR_OK; };
        if ((rc = up(s->d_spheres.p + b_sph * 4, &fl.spheres[b_sph * 4], (n_sph - b_sph) * 32)) || (rc = up(s->d_sphere_mat.p + b_sph, &fl.sphere_mat[b_sph], (n_sph - b_sph) * 4)) ||
            (rc = up(s->d_tri_v.p + b_tri * ZR_TRI_STRIDE, &fl.tri_v[b_tri * ZR_TRI_STRIDE], (n_tri - b_tri) * ZR_TRI_STRIDE * 8)) ||
            (rc = up(s->d_tri_s.p + b_tri * 20, &fl.tri_s[b_tri * 20], (n_tri - b_tri) * 160)) ||
            (rc = up(s->d_cubes.p + b_cube * 6, &fl.cubes[b_cube * 6], (n_cube - b_cube) * 48)) || (rc = up(s->d_cube_mat.p + b_cube, &fl.cube_mat[b_cube], (n_cube - b_cube) * 4)) ||
            (rc = up(s->d_media.p, fl.media.data(), n_media * sizeof(zr::DMedium))) || (rc = up(s->d_wrapped.p, fl.wrapped.data(), n_wrapped * sizeof(zr::DWrapped)))) return rc;
        HIP_OK(hipStreamSynchronize(st));   // (the staging arrays die with this block)
    }
    HIP_OK(hipStreamSynchronize(st));
    phase("node arrays + compound");
    cs.root = world.root; cs.quant_ok = world.quant_ok;
    for (const zr::BuiltTree& r : runs) cs.quant_ok = cs.quant_ok && r.quant_ok;
    cs.n_pairs = n_pairs; cs.n_quads = n_quads; cs.n_sph = n_sph; cs.n_tri = n_tri; cs.n_cube = n_cube; cs.n_pcube = n_pcube;
    cs.n_media = n_media; cs.n_wrapped = n_wrapped; cs.n_insts = n_insts; cs.plain_media = plain_media;
    cs.stack_demand = world.demand; cs.quad_depth = (int)world.quad_depth; cs.max_depth = (int)world.depth; cs.max_leaf = bp.max_leaf;
    for (int k = 0; k < 8; k++) cs.leaf_cnt[k] = cnt[k];
    cs.builder = "device (PLOC)";
    {   // the scratch arena, the trees' local records and the as-given copies: freed off the caller's clock
        struct Trash { std::shared_ptr<zr::DeviceBuilder> b; DevBuf<double> a0, a1, a2, a3, a4; DevBuf<uint32_t> u0, u1, u2, u3, u4, u5, u6; DevBuf<zr_medium> m; DevBuf<zr_object> o; DevBuf<uint8_t> c; int device; };
        auto t = std::make_shared<Trash>();
        t->device = s->ctx->device;
        t->b = std::move(builder);
        std::swap(t->a0, r_sph); std::swap(t->a1, r_tri_v); std::swap(t->a2, r_tri_n); std::swap(t->a3, r_cubes); std::swap(t->a4, r_gbox);
        std::swap(t->u0, r_sph_mat); std::swap(t->u1, r_tri_mat); std::swap(t->u2, r_cube_mat); std::swap(t->u3, d_inst_group); std::swap(t->u4, d_run_demand);
        std::swap(t->u5, d_run_root); std::swap(t->u6, d_run_qroot); std::swap(t->m, r_media); std::swap(t->o, r_objs); std::swap(t->c, r_code);
        try { std::thread([t]() mutable { (void)hipSetDevice(t->device); t.reset(); }).detach(); } catch (...) { /* no thread: freed here */ }
    }
    phase("release");
    return ZR_OK;
}

}  // namespace

extern "C" {

int zr_abi_version(void) { return ZR_ABI_VERSION; }
// helpers for zr_comm.cpp
int zr_internal_fail(int code, const char* msg) { return fail(code, "%s", msg); }
int zr_internal_device(const zr_ctx* c) { return c ? c->device : 0; }
const char* zr_last_error(void) { return g_err.c_str(); }

zr_ctx* zr_create(int device_ordinal) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(ZR_E_DEVICE, "no HIP device available (%s): this library has no CPU path", hipGetErrorString(e)); return nullptr; }
    if (device_ordinal < 0 || device_ordinal >= n) { fail(ZR_E_INVALID, "device ordinal %d out of range (0..%d)", device_ordinal, n - 1); return nullptr; }
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) { fail(ZR_E_DEVICE, "hipSetDevice: %s", hipGetErrorString(e)); return nullptr; }
    zr_ctx* c = new zr_ctx();
    c->device = device_ordinal;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(ZR_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); delete c; return nullptr;
    }
    for (int k = 1; k < ST_MAX_POOLS; k++)
        if ((e = hipStreamCreateWithFlags(&c->sub[k], hipStreamNonBlocking)) != hipSuccess) {
            fail(ZR_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); delete c; return nullptr;
        }
    if (c->d_ctr.alloc(16) != ZR_OK) { delete c; return nullptr; }
    c->variant = (int)env_double("ZR_KERNEL", 2);
    if (c->variant != 2) c->variant = 0;   // (variant 1, the round-1 wave-scheduler megakernel, is retired: 2-6 x slower and nothing depended on it)
    c->log_kind = (int)env_double("ZR_TIMELOG_KIND", 1);
    if (c->variant == 2) {
        c->st_blocks = zr::stream_extend_blocks();
        int over = (int)env_double("ZR_ST_BLOCKS", 0);
        if (over > 0) c->st_blocks = over;
        // up to 128 Mi slots = 26.8 GB of path state (of 288 GB), cut into two sub-pools of 64 Mi for worlds whose lean kernels run side by side (render_stream):
        // a launch that carries twice the rays walks the tree 9 % cheaper per ray (round 2: cfg3 368 ms with 64 Mi in one pool against 392 with 32 Mi, nothing more
        // from 96 or 128 Mi in ONE pool; round 4, two sub-pools: 351 / 340 / 341 ms with 64 / 96 / 128 Mi, profiles/r4_experiments_ab.txt).  A frame uses
        // min(this, units / 8) slots, so small frames and a rank's share of a sharded frame stay small
        c->st_slots = (uint32_t)env_double("ZR_STREAM_SLOTS", 128.0 * 1024 * 1024);
        // dynamic work units of shard s are handed out only by SHADE blocks with blockIdx % 64 == s (zr_stream.hip): a pool needs at
        // least 64 SHADE blocks (64 x 256 slots) or the units of the unserved shards would never be rendered
        c->st_slots = std::max<uint32_t>(64u * 256u, c->st_slots / 256 * 256);
        c->st_pools = (int)env_double("ZR_STREAM_POOLS", -1);
        if (env_double("ZR_STREAM_OVERLAP", -1) == 0) c->st_pools = 1;
        if (c->d_ctl.alloc((ST_MAX_POOLS + 1) * zr::stream_ctl_words()) != ZR_OK ||
            hipEventCreateWithFlags(&c->st_event, hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void**)&c->h_active, (ST_MAX_POOLS + 1) * zr::stream_ctl_words() * sizeof(unsigned int), 0) != hipSuccess) { fail(ZR_E_DEVICE, "variant-2 buffers: out of memory"); delete c; return nullptr; }
    }
    return c;
}

void zr_destroy(zr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    c->d_ctr.release(); c->d_out.release(); c->d_tiles.release();
    c->d_pool.release(); c->d_pixels.release(); c->d_partial.release(); c->d_kend.release(); c->d_cls.release(); c->d_cpart.release(); c->d_ctl.release(); c->d_st_overflow.release();
    if (c->h_active) (void)hipHostFree(c->h_active);
    if (c->st_event) (void)hipEventDestroy(c->st_event);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (int k = 1; k < ST_MAX_POOLS; k++) if (c->sub[k]) (void)hipStreamDestroy(c->sub[k]);
    delete c;
}

zr_scene* zr_scene_create(zr_ctx* c) {
    if (!c) { fail(ZR_E_INVALID, "null context"); return nullptr; }
    zr_scene* s = new zr_scene();
    s->ctx = c;
    return s;
}
void zr_scene_destroy(zr_scene* s) {
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    delete s;
}

// A borrowed commit drops every geometry view (`released`): the scene can be committed again only after ALL of them were given
// again — zr_scene_set_all / zr_scene_set_all_borrowed, or each of the six geometry setters (SET_* bits).  A lone
// zr_scene_set_materials must not re-arm the commit: it would build an empty world and return ZR_OK.
enum { SET_SPHERES = 1, SET_TRIANGLES = 2, SET_CUBES = 4, SET_MEDIA = 8, SET_OPS = 16, SET_OBJECTS = 32, SET_ALL_GEOMETRY = 63 };
#define CHECK_SCENE(s) do { if (!(s)) return fail(ZR_E_INVALID, "null scene"); (s)->committed = false; } while (0)
#define GEOMETRY_SET(s, bit) do { if ((s)->released) { (s)->reset_mask |= (bit); if (((s)->reset_mask & SET_ALL_GEOMETRY) == SET_ALL_GEOMETRY) { (s)->released = false; (s)->reset_mask = 0; } } } while (0)

int zr_scene_set_spheres(zr_scene* s, const double* p, const uint32_t* mat, size_t n) {
    CHECK_SCENE(s);
    if (n && (!p || !mat)) return fail(ZR_E_INVALID, "null sphere arrays");
    s->spheres.copy(p, n * 4); s->sphere_mat.copy(mat, n);
    GEOMETRY_SET(s, SET_SPHERES);
    return ZR_OK;
}
int zr_scene_set_triangles(zr_scene* s, const double* v9, const double* n9, const uint32_t* mat, size_t n) {
    CHECK_SCENE(s);
    if (n && (!v9 || !n9 || !mat)) return fail(ZR_E_INVALID, "null triangle arrays");
    s->tri_v.copy(v9, n * 9); s->tri_n.copy(n9, n * 9); s->tri_mat.copy(mat, n);
    GEOMETRY_SET(s, SET_TRIANGLES);
    return ZR_OK;
}
int zr_scene_set_cubes(zr_scene* s, const double* q, const uint32_t* mat, size_t n) {
    CHECK_SCENE(s);
    if (n && (!q || !mat)) return fail(ZR_E_INVALID, "null cube arrays");
    s->cubes.copy(q, n * 12); s->cube_mat.copy(mat, n);
    GEOMETRY_SET(s, SET_CUBES);
    return ZR_OK;
}
int zr_scene_set_media(zr_scene* s, const zr_medium* m, size_t n) {
    CHECK_SCENE(s);
    if (n && !m) return fail(ZR_E_INVALID, "null media array");
    s->media.copy(m, n);
    GEOMETRY_SET(s, SET_MEDIA);
    return ZR_OK;
}
int zr_scene_set_xform_ops(zr_scene* s, const zr_xform_op* o, size_t n) {
    CHECK_SCENE(s);
    if (n && !o) return fail(ZR_E_INVALID, "null op array");
    s->ops.copy(o, n);
    GEOMETRY_SET(s, SET_OPS);
    return ZR_OK;
}
int zr_scene_set_objects(zr_scene* s, const zr_object* o, size_t n) {
    CHECK_SCENE(s);
    if (n && !o) return fail(ZR_E_INVALID, "null object array");
    s->objects.copy(o, n); s->objects_set = n > 0;
    GEOMETRY_SET(s, SET_OBJECTS);
    return ZR_OK;
}
int zr_scene_set_groups(zr_scene* s, const zr_group* g, size_t n) {
    CHECK_SCENE(s);
    if (n && !g) return fail(ZR_E_INVALID, "null group array");
    s->groups.assign(g, g + n);
    return ZR_OK;
}
int zr_scene_set_materials(zr_scene* s, const zr_material* m, size_t n) {
    CHECK_SCENE(s);
    if (n && !m) return fail(ZR_E_INVALID, "null material array");
    s->materials.assign(m, m + n);
    return ZR_OK;
}
int zr_scene_set_textures(zr_scene* s, const zr_texture* t, size_t n, const void* blob, size_t bytes) {
    CHECK_SCENE(s);
    if ((n && !t) || (bytes && !blob)) return fail(ZR_E_INVALID, "null texture arrays");
    s->textures.assign(t, t + n);
    s->texels.copy((const unsigned char*)blob, bytes);
    return ZR_OK;
}
int zr_scene_set_all(zr_scene* s, const zr_scene_desc* d) {
    CHECK_SCENE(s);
    if (!d) return fail(ZR_E_INVALID, "null scene description");
    int rc;
    if ((rc = zr_scene_set_spheres(s, d->spheres, d->sphere_mat, d->n_spheres))) return rc;
    if ((rc = zr_scene_set_triangles(s, d->tri_v, d->tri_n, d->tri_mat, d->n_tris))) return rc;
    if ((rc = zr_scene_set_cubes(s, d->cubes, d->cube_mat, d->n_cubes))) return rc;
    if ((rc = zr_scene_set_media(s, d->media, d->n_media))) return rc;
    if ((rc = zr_scene_set_xform_ops(s, d->ops, d->n_ops))) return rc;
    if ((rc = zr_scene_set_objects(s, d->objects, d->n_objects))) return rc;
    if ((rc = zr_scene_set_groups(s, d->groups, d->n_groups))) return rc;
    if ((rc = zr_scene_set_materials(s, d->materials, d->n_materials))) return rc;
    return zr_scene_set_textures(s, d->textures, d->n_textures, d->texels, d->texel_bytes);
}

int zr_scene_set_all_borrowed(zr_scene* s, const zr_scene_desc* d) {
    CHECK_SCENE(s);
    if (!d) return fail(ZR_E_INVALID, "null scene description");
    if ((d->n_spheres && (!d->spheres || !d->sphere_mat)) || (d->n_tris && (!d->tri_v || !d->tri_n || !d->tri_mat)) || (d->n_cubes && (!d->cubes || !d->cube_mat)) ||
        (d->n_media && !d->media) || (d->n_ops && !d->ops) || (d->n_objects && !d->objects) || (d->texel_bytes && !d->texels))
        return fail(ZR_E_INVALID, "null array in the scene description");
    s->spheres.borrow(d->spheres, d->n_spheres * 4); s->sphere_mat.borrow(d->sphere_mat, d->n_spheres);
    s->tri_v.borrow(d->tri_v, d->n_tris * 9); s->tri_n.borrow(d->tri_n, d->n_tris * 9); s->tri_mat.borrow(d->tri_mat, d->n_tris);
    s->cubes.borrow(d->cubes, d->n_cubes * 12); s->cube_mat.borrow(d->cube_mat, d->n_cubes);
    s->media.borrow(d->media, d->n_media);
    s->ops.borrow(d->ops, d->n_ops);
    s->objects.borrow(d->objects, d->n_objects); s->objects_set = d->n_objects > 0;
    s->texels.borrow((const unsigned char*)d->texels, d->texel_bytes);
    s->borrowed = true; s->released = false; s->reset_mask = 0;
    int rc;
    if ((rc = zr_scene_set_groups(s, d->groups, d->n_groups))) return rc;
    if ((rc = zr_scene_set_materials(s, d->materials, d->n_materials))) return rc;   // the small tables are copied: render calls validate against them
    if (d->n_textures && !d->textures) return fail(ZR_E_INVALID, "null texture array");
    s->textures.assign(d->textures, d->textures + d->n_textures);
    return ZR_OK;
}

int zr_scene_commit(zr_scene* s) {
    if (!s) return fail(ZR_E_INVALID, "null scene");
    if (s->released) return fail(ZR_E_STATE, "the arrays given to zr_scene_set_all_borrowed were released by the previous commit: set the scene again");
    s->committed = false;
    HIP_OK(hipSetDevice(s->ctx->device));
    // the world list
    std::vector<zr_object> objs;
    if (s->objects_set) objs.assign(s->objects.begin(), s->objects.end());
    else {
        std::vector<char> sb(s->sphere_mat.size(), 0), cb(s->cube_mat.size(), 0);
        for (const zr_medium& m : s->media) {
            if (m.boundary_type == ZR_PRIM_SPHERE && m.boundary_index < sb.size()) sb[m.boundary_index] = 1;
            if (m.boundary_type == ZR_PRIM_CUBE && m.boundary_index < cb.size()) cb[m.boundary_index] = 1;
        }
        for (uint32_t k = 0; k < s->sphere_mat.size(); k++) if (!sb[k]) objs.push_back({ZR_PRIM_SPHERE, k, 0, 0});
        for (uint32_t k = 0; k < s->tri_mat.size(); k++) objs.push_back({ZR_PRIM_TRIANGLE, k, 0, 0});
        for (uint32_t k = 0; k < s->cube_mat.size(); k++) if (!cb[k]) objs.push_back({ZR_PRIM_CUBE, k, 0, 0});
        for (uint32_t k = 0; k < s->media.size(); k++) objs.push_back({ZR_PRIM_MEDIUM, k, 0, 0});
    }
    const bool commit_stats = std::getenv("ZR_COMMIT_STATS") != nullptr;
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_phase = now_s();
    auto phase = [&](const char* what) { if (commit_stats) { const double t = now_s(); std::fprintf(stderr, "[zr] commit: %-22s %.1f ms\n", what, (t - t_phase) * 1e3); t_phase = t; } };
    int rc = validate(*s, objs);
    if (rc) return rc;
    phase("world list + validate");
    if (s->media.size() > 65535) return fail(ZR_E_INVALID, "at most 65535 media (RNG key layout, zr_rng.h)");
    {   // which builder.  ZR_BVH_BUILD=device | host forces one; otherwise worlds of at least ZR_BVH_DEVICE_MIN entries (131072: from
        // there on the device build's top is arranged by SAH, zr_build.h) are built on the device: cfg3's 1M triangles commit in
        // 19 ms instead of 106 and the frame takes 1 % longer than on the host's tree (EXTEND alone 2.6 %) — profiles/r3_builders.txt;
        // a small world is built faster by the host than a few dozen kernel launches take
        const char* bm = std::getenv("ZR_BVH_BUILD");
        const bool force_dev = bm && std::strcmp(bm, "device") == 0, force_host = bm && std::strcmp(bm, "host") == 0;
        const bool use_dev = !force_host && !objs.empty() && objs.size() < (1u << 30) && (force_dev || (double)objs.size() >= env_double("ZR_BVH_DEVICE_MIN", 131072));
        if (use_dev) {
            CommitSummary cs;
            rc = commit_device(s, objs, commit_stats, cs);
            if (rc == ZR_OK) return finish_commit(s, cs, objs.size());
            if (rc != ZR_FALLBACK_HOST) return rc;
            phase("device build refused");
        }
    }

    // two-level BVH: every group of triangles gets a tree of its own, in its own space, once — however many objects place it
    std::vector<zr::BuildResult> runs(s->groups.size());
    std::vector<zr::BuildBox> group_box(s->groups.size());
    {
        Boxer tri_boxer{*s};
        const double ck_tri[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        for (size_t g = 0; g < s->groups.size(); g++) {
            const zr_group& grp = s->groups[g];
            std::vector<zr::BuildBox> tb(grp.triangle_count);
            std::vector<uint32_t> tk(grp.triangle_count, ZR_PRIM_TRIANGLE);
            zr::BuildBox all; for (int a = 0; a < 3; a++) { all.lo[a] = kInf; all.hi[a] = -kInf; }
            for (uint32_t k = 0; k < grp.triangle_count; k++) {
                tb[k] = tri_boxer.prim(ZR_PRIM_TRIANGLE, grp.first_triangle + k);
                for (int a = 0; a < 3; a++) { all.lo[a] = std::fmin(all.lo[a], tb[k].lo[a]); all.hi[a] = std::fmax(all.hi[a], tb[k].hi[a]); }
            }
            group_box[g] = all;
            zr::build_bvh(tb, tk, 4, ZR_STACK_DEPTH - 2, 1.0, ck_tri, runs[g]);
            if (runs[g].max_depth >= ZR_STACK_DEPTH - 1) return fail(ZR_E_INVALID, "group %zu: BVH depth %d exceeds the traversal stack", g, runs[g].max_depth);
        }
    }
    // boxes + kinds
    Boxer boxer{*s, &group_box};
    std::vector<zr::BuildBox> boxes(objs.size());
    std::vector<uint32_t> kinds(objs.size());
    std::vector<uint8_t> baked(objs.size(), 0);
    const bool bake = env_double("ZR_BAKE_TRIANGLES", 1) != 0;
    std::atomic<size_t> bad_box{(size_t)-1};
    {
        unsigned hw = std::thread::hardware_concurrency();
        const size_t nobj = objs.size();
        const int T = nobj < 65536 ? 1 : (int)std::max(1u, std::min(16u, hw));
        auto work = [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++) {
            const zr_object& o = objs[k];
            boxes[k] = boxer.chain(o.type, o.index, o.chain_first, o.chain_count);
            classify_object(*s, o, bake, kinds[k], baked[k]);
            for (int a = 0; a < 3; a++)
                if (!std::isfinite(boxes[k].lo[a]) || !std::isfinite(boxes[k].hi[a])) { size_t want = (size_t)-1; bad_box.compare_exchange_strong(want, k); }
        }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, nobj * t / T, nobj * (t + 1) / T);
        work(0, nobj / T);
        for (auto& x : th) x.join();
    }
    if (bad_box.load() != (size_t)-1) return fail(ZR_E_INVALID, "object %zu has a non-finite bounding box", bad_box.load());
    zr::BuildResult br;
    double ck[8] = {env_double("ZR_BVH_COST_SPHERE", 1.0), env_double("ZR_BVH_COST_TRI", 1.5), env_double("ZR_BVH_COST_CUBE", 1.0),
                    env_double("ZR_BVH_COST_MEDIUM", 3.0), env_double("ZR_BVH_COST_WRAPPED", 3.0), env_double("ZR_BVH_COST_PCUBE", 1.5),
                    env_double("ZR_BVH_COST_GROUP", 16.0), 1};
    int max_leaf = (int)env_double("ZR_BVH_MAX_LEAF", 4);
    // cubes, media and wrapped objects are few, large and dear to test: one per leaf, so that a ray only tests those whose own box it enters
    const int big = (int)env_double("ZR_BVH_MAX_LEAF_BIG", 1);
    const int leaf_cap[8] = {0, 0, big, big, big, big, 1, 0};   // a placement is always a leaf of its own (EXTEND enters it as a whole)
    phase("boxes");
    zr::build_bvh(boxes, kinds, max_leaf, ZR_STACK_DEPTH - 2, env_double("ZR_BVH_COST_TRAVERSE", 1.0), ck, br, leaf_cap);
    phase("binned-SAH build");
    if (br.max_depth >= ZR_STACK_DEPTH - 1) return fail(ZR_E_INVALID, "BVH depth %d exceeds the traversal stack", br.max_depth);

    // the flattener lives on the heap: it is handed, with everything else that is large, to a thread that frees it (below)
    std::shared_ptr<Flattener> flp(new Flattener{*s, objs, br});
    Flattener& fl = *flp;
    fl.baked = &baked;
    fl.runs = runs.empty() ? nullptr : &runs;
    fl.open_ratio = env_double("ZR_BVH_OPEN_RATIO", 1.25);
    // the primitive arrays (a quarter of a gigabyte for a million triangles) go to the device while the host still plans and
    // numbers the 4-wide nodes: a thread of its own, joined before the node arrays follow
    int up_rc = ZR_OK;
    std::string up_err;
    std::thread uploader;
    const int device = s->ctx ? s->ctx->device : 0;
    fl.after_primitives = [&]() {
        uploader = std::thread([&, device]() {
            auto go = [&]() -> int {
                HIP_OK(hipSetDevice(device));
                int r;
                if ((r = s->d_spheres.upload(fl.spheres))) return r;
                if ((r = s->d_sphere_mat.upload(fl.sphere_mat))) return r;
                if ((r = s->d_tri_v.upload(fl.tri_v))) return r;
                if ((r = s->d_tri_s.upload(fl.tri_s))) return r;
                if ((r = s->d_cubes.upload(fl.cubes))) return r;
                if ((r = s->d_cube_mat.upload(fl.cube_mat))) return r;
                if ((r = s->d_pcubes.upload(fl.pcubes))) return r;
                if ((r = s->d_pcube_mat.upload(fl.pcube_mat))) return r;
                if ((r = s->d_media.upload(fl.media))) return r;
                if ((r = s->d_wrapped.upload(fl.wrapped))) return r;
                return ZR_OK;
            };
            up_rc = go();
            if (up_rc != ZR_OK) up_err = g_err;   // the error text is per thread
        });
    };
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};   // no path leaves the thread running
    fl.run();
    phase("flatten + quantise");

    if ((rc = s->d_nodes.upload(fl.pairs))) return rc;
    if ((rc = s->d_quads.upload(fl.quads))) return rc;
    if ((rc = s->d_insts.upload(fl.insts))) return rc;   // (after the groups' nodes were numbered: a placement names its group's root)
    if (uploader.joinable()) uploader.join();
    else if (fl.after_primitives) {   // a world without nodes returned from run() before the hook: upload the (empty) arrays here
        fl.after_primitives(); if (uploader.joinable()) uploader.join();
    }
    if (up_rc != ZR_OK) return fail(up_rc, "%s", up_err.c_str());
    CommitSummary cs;
    cs.root = fl.root; cs.quant_ok = fl.quant_ok; cs.n_pairs = fl.pairs.size(); cs.n_quads = fl.quads.size();
    cs.n_sph = fl.sphere_mat.size(); cs.n_tri = fl.tri_s.size() / 20; cs.n_cube = fl.cube_mat.size(); cs.n_pcube = fl.pcube_mat.size();
    cs.n_media = fl.media.size(); cs.n_wrapped = fl.wrapped.size(); cs.n_insts = fl.insts.size();
    cs.plain_media = true;   // media whose boundary is an unwrapped sphere or cube
    for (size_t k = 0; k < fl.media.size(); k++) if (fl.media[k].chain_count != 0) cs.plain_media = false;
    cs.stack_demand = fl.stack_demand(); cs.quad_depth = fl.quad_depth; cs.max_depth = br.max_depth; cs.max_leaf = max_leaf; cs.kept_closed = fl.n_kept_closed;
    for (int k = 0; k < 8; k++) cs.leaf_cnt[k] = fl.cnt[k];
    cs.builder = "host (binned SAH)";
    if ((rc = finish_commit(s, cs, objs.size()))) return rc;
    phase("upload");
    {   // unmapping half a gigabyte of staging arrays takes tens of milliseconds: not on the caller's clock
        struct Trash { std::shared_ptr<Flattener> fl; zr::BuildResult br; std::vector<zr::BuildBox> boxes; std::vector<zr_object> objs;
                       std::vector<uint32_t> kinds; std::vector<uint8_t> baked; std::vector<zr::BuildResult> runs; };
        auto t = std::make_shared<Trash>();
        fl.after_primitives = nullptr;   // (it captures locals of this call)
        t->fl = std::move(flp); t->br = std::move(br); t->boxes = std::move(boxes); t->objs = std::move(objs); t->kinds = std::move(kinds); t->baked = std::move(baked); t->runs = std::move(runs);
        try { std::thread([t]() mutable { t.reset(); }).detach(); } catch (...) { /* no thread: freed here */ }
    }
    phase("release");
    return ZR_OK;
}

int zr_scene_stats(const zr_scene* s, uint64_t out[4]) {
    if (!s || !s->committed) return fail(ZR_E_STATE, "scene not committed");
    std::memcpy(out, s->stats, sizeof s->stats);
    return ZR_OK;
}

uint32_t zr_scene_traversal_stack(const zr_scene* s) { return s && s->committed ? s->stack_demand : 0u; }
const char* zr_scene_builder(const zr_scene* s) { return s && s->committed ? s->builder : ""; }

}  // extern "C"

namespace {

// camera::initialize, camera.hpp:358-399
void make_camera(const zr_camera& c, zr::DCamera& d) {
    int W = c.image_width < 1 ? 1 : c.image_width, H = c.image_height < 1 ? 1 : c.image_height;
    double aspect = double(W) / H;
    H3 center = h3(c.lookfrom), lookat = h3(c.lookat), vup = h3(c.vup);
    double theta = c.vfov * kPi / 180.0;
    double h = std::tan(theta / 2);
    double vh = 2 * h * c.focus_dist;
    double vw = vh * aspect;
    H3 w = unit(center - lookat);
    H3 u = unit(cross(vup, w));
    H3 v = cross(w, u);
    H3 vu = vw * u;
    H3 vv = vh * -v;
    H3 du = vu / W;
    H3 dv = vv / H;
    H3 ul = center - (c.focus_dist * w) - vu / 2 - vv / 2;
    H3 p00 = ul + 0.5 * (du + dv);
    double rad = c.focus_dist * std::tan((c.defocus_angle / 2) * kPi / 180.0);
    st3(d.center, center); st3(d.pixel00, p00); st3(d.du, du); st3(d.dv, dv);
    st3(d.disk_u, u * rad); st3(d.disk_v, v * rad);
    d.W = W; d.H = H; d.spp = c.samples_per_pixel < 1 ? 1 : c.samples_per_pixel; d.max_depth = c.max_depth;
    d.defocus = !(c.defocus_angle <= 0) ? 1 : 0;
    d.pad_ = 0;
}

// ray-independent part of get_background_color, camera.hpp:832-834, 844-858, 874-895, 914-918
void make_env(const zr_env& e, zr::DEnv& d) {
    std::memset(&d, 0, sizeof d);
    d.mode = e.mode; d.hdr_tex = e.hdr_texture; d.intensity = e.intensity;
    H3 bg = h3(e.background_color) * e.intensity;
    st3(d.solid, bg);
    d.cy = std::cos(e.hdri_rotation); d.sy = std::sin(e.hdri_rotation);
    d.cp = std::cos(e.hdri_tilt); d.sp = std::sin(e.hdri_tilt);
    d.cr = std::cos(e.hdri_roll); d.sr = std::sin(e.hdri_roll);
    H3 sun = unit(h3(e.sun_direction));
    double sh = sun.y;
    double ah = sh - 0.05;
    double sky_exposure = clampd(ah * 8.0 + 1.4, 0.0, 1.0);
    double day = clampd(ah * 10.0 + 1.1, 0.0, 1.0);
    double sunset_i = clampd(1.0 - std::fabs(ah + 0.05) * 30.0, 0.0, 1.0);
    double sunset = (ah > -0.1) ? sunset_i : 0.0;
    if (sh < 0) sunset *= (sh * 10.0 + 1.0);
    sunset = clampd(sunset, 0.0, 1.0);
    H3 zen = H3{0.01, 0.03, 0.1} * (1.0 - day) + H3{0.2, 0.5, 1.0} * day;
    H3 hor = H3{0.05, 0.02, 0.01} * (1.0 - day) + H3{0.6, 0.8, 1.0} * day;
    hor = hor * (1.0 - sunset) + H3{1.0, 0.35, 0.1} * sunset;
    st3(d.sun, sun); st3(d.horizon, hor); st3(d.zenith, zen);
    d.sky_scale = e.intensity * 1.5; d.sky_exposure = sky_exposure;
    d.sun_thr = 1.0 - (e.sun_size * 0.001);
    d.sun_on = ah > -0.1 ? 1 : 0;
    H3 scol = h3(e.sun_color) * (1.0 - sunset) + H3{1.0, 0.3, 0.1} * sunset;
    double vis = clampd(sh * 5.0 + 1.0, 0.0, 1.0);
    st3(d.sun_add, (scol * e.sun_intensity) * vis);
}

struct Plan {
    int W, H, ts, tiles_x, tiles_y, x0, y0, x1, y1, lanes;
    std::vector<int32_t> tiles;
};

int make_plan(const zr_camera& cam, const zr_region* region, Plan& p) {
    p.W = cam.image_width < 1 ? 1 : cam.image_width;
    p.H = cam.image_height < 1 ? 1 : cam.image_height;
    p.ts = 32; int mod = 1, rem = 0;
    p.x0 = 0; p.y0 = 0; p.x1 = p.W; p.y1 = p.H;
    if (region) {
        if (region->tile_size > 0) p.ts = region->tile_size;
        if (region->tile_mod > 1) { mod = region->tile_mod; rem = region->tile_rem; }
        if (region->w > 0 && region->h > 0) { p.x0 = region->x0; p.y0 = region->y0; p.x1 = region->x0 + region->w; p.y1 = region->y0 + region->h; }
    }
    if (p.x0 < 0 || p.y0 < 0 || p.x1 > p.W || p.y1 > p.H || rem < 0 || rem >= mod || p.ts > 1024)
        return fail(ZR_E_INVALID, "region outside the %dx%d frame or bad tile parameters", p.W, p.H);
    p.tiles_x = (p.W + p.ts - 1) / p.ts; p.tiles_y = (p.H + p.ts - 1) / p.ts;
    p.tiles.clear();
    for (int ty = p.y0 / p.ts; ty <= (p.y1 - 1) / p.ts; ty++)
        for (int tx = p.x0 / p.ts; tx <= (p.x1 - 1) / p.ts; tx++) {
            int t = ty * p.tiles_x + tx;
            if (t % mod == rem) p.tiles.push_back(t);
        }
    int spp = cam.samples_per_pixel < 1 ? 1 : cam.samples_per_pixel;
    p.lanes = 64; while (p.lanes > spp) p.lanes >>= 1;
    return ZR_OK;
}

int resolve_times(zr_ctx* c);

// spill slabs of the EXTEND traversal stack, sized for the deepest tree this context has met (never below 36 levels, the
// fixed size of round 1): one slab per resident wave and sub-pool
int ensure_stack_slabs(zr_ctx* c, const zr_scene* s) {
    const uint32_t need = std::max<uint32_t>(36u, zr::stream_overflow_levels(s->stack_demand));
    if (need <= c->st_ovf_levels && c->d_st_overflow.p) return ZR_OK;
    HIP_OK(hipDeviceSynchronize());
    c->d_st_overflow.release();
    int rc = c->d_st_overflow.alloc(ST_MAX_POOLS * zr::stream_overflow_bytes(c->st_blocks, need));
    if (rc) { c->st_ovf_levels = 0; return rc; }
    c->st_ovf_levels = need;
    return ZR_OK;
}

struct HostTimer : zr::StreamTimer {
    zr_ctx* c; hipEvent_t cur_a = nullptr; bool ok = true;
    explicit HostTimer(zr_ctx* c) : c(c) {}
    hipEvent_t get() {
        if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { ok = false; return nullptr; }
        return e;
    }
    void begin(hipStream_t st, int) override { cur_a = get(); if (cur_a) (void)hipEventRecord(cur_a, st); }
    void end(hipStream_t st, int kind) override {
        hipEvent_t b = get();
        if (!cur_a || !b) return;
        (void)hipEventRecord(b, st);
        zr_ctx::Pending pe{}; pe.a = cur_a; pe.b = b; pe.render_id = c->render_id; pe.kind = kind;
        c->pending.push_back(pe);
        cur_a = nullptr;
    }
};

// variant 2: streaming wavefront pipeline (zr_stream.hip).  Synchronises the stream internally (the round loop
// needs the active-slot count), so zr_render_device returns with the frame complete.
// mode 0: the render; 1 / 2: beauty pass and replay pass of the reflection / refraction split (zr_stream.hip, stream_shade)
int render_stream(zr_ctx* c, const zr_scene* s, const zr::DCamera& dc, const zr::DEnv& de, uint64_t seed, const Plan& plan, int count,
                  double* d_out, hipStream_t stream, volatile const uint8_t* keep_going, int mode = 0, double* d_out2 = nullptr,
                  zr::StreamProgress* progress = nullptr) {
    // pixel list (cached per plan)
    std::vector<int32_t> key = {plan.W, plan.H, plan.ts, plan.x0, plan.y0, plan.x1, plan.y1, (int32_t)plan.tiles.size(),
                                plan.tiles.empty() ? -1 : plan.tiles.front(), plan.tiles.empty() ? -1 : plan.tiles.back(),
                                (int32_t)env_double("ZR_STREAM_BOTTOM_UP", 1)};
    if (plan.W > 65535 || plan.H > 65535) return fail(ZR_E_INVALID, "kernel variant 2 supports frames up to 65535 x 65535");
    if (key != c->pix_key || !c->d_pixels.p) {
        std::vector<uint32_t> pix;
        pix.reserve((size_t)plan.tiles.size() * plan.ts * plan.ts);
        for (int32_t t : plan.tiles) {
            int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
            int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
            for (int y = ya; y < yb; y++) for (int x = xa; x < xb; x++) pix.push_back((uint32_t)x | ((uint32_t)y << 16));
        }
        // Work units are handed out in pixel-list order, and when they run out the frame DRAINS: the paths still alive need up to
        // max_depth more rounds, each with fewer rays than the chip wants (10 rounds = 16 ms of a 415 ms cfg3 frame, 10 of the
        // 60 ms of a rank's 1/8 share).  The drain is as long as the paths started last, so the list runs BOTTOM-UP: the top
        // of a frame is where the sky is, and a sky sample ends in one round.  The image does not depend on the order (every
        // sample is written once and reduced in a fixed order).
        if (env_double("ZR_STREAM_BOTTOM_UP", 1) != 0) std::reverse(pix.begin(), pix.end());
        int rc = c->d_pixels.upload(pix);
        if (rc) return rc;
        c->pix_key = key;
    }
    const uint32_t n_pix = (uint32_t)c->d_pixels.n;
    if (c->pending.size() > 65536) { int rr = resolve_times(c); if (rr) return rr; }
    c->render_id++; c->last_stream = stream; c->last_counted = count != 0; c->last_rounds = 0;
    HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 16 * sizeof(unsigned long long), stream));
    if (n_pix == 0) return ZR_OK;
    const uint32_t spp = (uint32_t)dc.spp;
    const uint64_t units = (uint64_t)n_pix * spp;   // one work unit per primary sample
    if (units > 0xFFFFFFFFull) return fail(ZR_E_INVALID, "frame too large for kernel variant 2 (pixels x spp must fit 32 bits); shard it (zr_region) or set ZR_KERNEL=0");
    // slot pool: large enough to fill the chip every round, small enough that the frame takes dozens of rounds (a
    // rank that owns 1/8 of the tiles must not degenerate into one shrinking batch)
    int rc;
    // per-sample radiance first: without it this pipeline cannot run at all (the caller falls back to the pixel-group kernel)
    const size_t samples_n = (size_t)units * 3;
    if (c->d_partial.n < samples_n) {
        HIP_OK(hipStreamSynchronize(stream));
        if (c->d_partial.alloc(samples_n) != ZR_OK) return fail(ZR_E_NOMEM, "no device memory for the per-sample radiance buffer (%zu bytes)", samples_n * sizeof(double));
    }
    // A world of a handful of objects is rendered by the FUSED kernel (zr_stream.hip: fused_render): every object tested per
    // segment, the path in registers, no tree, no slot pool.  Testing all objects costs time in proportion to their number, the
    // pipeline about the same per segment whatever the scene: the switch-over is ZR_FUSED_MAX objects.  A caller that polls
    // (cancellation, lines_rendered, live preview) gets the frame in sixteen launches with the poll between them; the split passes
    // stay on the pipeline.
    if (mode == 0 && s->fused_ok && s->leaf_level <= 2 && s->leaf_objects > 0 && (double)s->leaf_objects <= env_double("ZR_FUSED_MAX", ZR_FUSED_OBJECTS) &&
        env_double("ZR_FUSED", 1) != 0) {
        if (c->fused_blocks == 0) c->fused_blocks = zr::fused_blocks();
        HostTimer ftimer(c);
        int parts = 1;
        if (keep_going || progress) HIP_OK(hipMemsetAsync(c->d_partial.p, 0, samples_n * sizeof(double), stream));   // a cancelled frame / a preview reduces what exists
        hipError_t fe = zr::fused_render_frame(s->ds, dc, de, seed, spp, n_pix, c->d_pixels.p, c->d_partial.p, c->d_ctl.p, c->fused_blocks, d_out, c->d_ctr.p, count != 0,
                                               s->leaf_level <= 1 ? 1 : 2, stream, &ftimer, s->fused, keep_going, progress, &parts);
        if (fe != hipSuccess) return fail(ZR_E_DEVICE, "fused small-scene kernel failed: %s", hipGetErrorString(fe));
        c->last_rounds = (uint64_t)(parts < 0 ? -parts : parts); c->last_path = 3;
        HIP_OK(hipStreamSynchronize(stream));
        if (parts < 0) return fail(ZR_E_CANCELLED, "render cancelled after %d of 16 parts", -parts);
        return ZR_OK;
    }
    c->last_path = 2;
    if ((rc = ensure_stack_slabs(c, s))) return rc;
    // slot pool: large enough to fill the chip every round, small enough that the frame takes dozens of rounds (a
    // rank that owns 1/8 of the tiles must not degenerate into one shrinking batch)
    const bool affine = env_double("ZR_STREAM_AFFINE", 0) != 0;   // measured: -18 % L2 requests, -14 % misses, frame time +1 % (profiles/r3_affine_ab.txt): off
    uint32_t P = 0, unit_chunk = 0, drain_slots = 0;
    size_t drain_at = 0;
    for (uint32_t cap = c->st_slots;; cap /= 2) {
        P = cap / 64 * 64;
        uint64_t want = std::max<uint64_t>(units / (uint64_t)std::max(1.0, env_double("ZR_STREAM_UNITS_PER_SLOT", 8)), 1u << 20);
        want = want / 64 * 64;
        if (want < P) P = (uint32_t)want;
        if (units < P) P = (uint32_t)((units + 63) / 64 * 64);
        // XCD-affine hand-out of the work units (zr_stream.hip: st_unit_of): chunks of `unit_chunk` units — by default the samples of
        // 1024 consecutive pixels of the tile-ordered list, i.e. one 32 x 32 tile — belong to one shard, hence to one XCD's L2
        unit_chunk = 0;
        if (affine) {
            const uint32_t round = 64u * 256u;   // a pool is whole rounds of ST_SHARDS SHADE blocks
            P = std::max<uint32_t>(round, P / round * round);
            if (units < P) P = (uint32_t)((units + round - 1) / round * round);
            uint64_t G = (uint64_t)std::max(1.0, env_double("ZR_STREAM_CHUNK_PX", 1024)) * spp;
            G = std::min<uint64_t>(G, units / (64u * 8u));   // every shard gets at least eight chunks (small frames: smaller chunks)
            unit_chunk = (uint32_t)std::max<uint64_t>(256, std::min<uint64_t>(G, 1u << 30));
        }
        // the slot pool, its sub-pools' rounding, and behind them the small pool the survivors of a frame's drain are moved to
        // (zr_stream.hip: stream_compact).  Sized for the P this frame uses and only ever grown: a 64 x 64 test frame or a one-ray
        // device_hit() does not reserve the 13.7 GB a 1080p frame at 512 spp wants ("Device memory")
        drain_slots = P / 16 / 256 * 256 + 256;
        drain_at = zr::stream_pool_bytes(P) + 65536 * ST_MAX_POOLS;
        const size_t pool_need = drain_at + zr::stream_pool_bytes(drain_slots);
        if (c->d_pool.n >= pool_need) break;
        HIP_OK(hipStreamSynchronize(stream));
        if (c->d_pool.alloc(pool_need) == ZR_OK) break;
        // a pool that cannot be had is retried at half the size: the frame takes more rounds, the image is the same
        if (cap <= (1u << 20)) return fail(ZR_E_NOMEM, "no device memory for a slot pool of %u paths (%zu bytes)", P, pool_need);
        std::fprintf(stderr, "[zr] no device memory for a pool of %u path slots (%zu bytes): retrying with half\n", P, pool_need);
    }
    const bool use_drain = env_double("ZR_STREAM_DRAIN_POOL", 1) != 0;
    if (keep_going || progress) HIP_OK(hipMemsetAsync(c->d_partial.p, 0, samples_n * sizeof(double), stream));  // a cancelled frame / a preview reduces what exists
    if (mode != 0) {
        if (c->d_kend.n < units * 2) { if ((rc = c->d_kend.alloc(units * 2))) return rc; }
        if (c->d_cls.n < units) { if ((rc = c->d_cls.alloc(units))) return rc; }
        if (mode == 2) HIP_OK(hipMemsetAsync(c->d_cls.p, 0, units, stream));
        const size_t cp = ((size_t)c->st_slots / 256 + ST_MAX_POOLS + 1) * 4;
        if (c->d_cpart.n < cp) { if ((rc = c->d_cpart.alloc(cp))) return rc; }
    }
    HostTimer timer(c);
    int rounds = 0;
    hipStream_t streams[ST_MAX_POOLS];
    streams[0] = stream;
    for (int k = 1; k < ST_MAX_POOLS; k++) streams[k] = c->sub[k];
    const bool sharded = (size_t)plan.tiles.size() < (size_t)plan.tiles_x * plan.tiles_y;
    // Two sub-pools, a fraction of a round apart on two streams, let one pool's SHADE run beside the other's EXTEND.  Until round 3 that paid on a rank's share only
    // (the whole frame: 366.6 against 365.7 ms): SHADE needed 124 registers and found no room beside EXTEND's waves.  The lean builds of both kernels use 80
    // (zr_stream.hip), a SIMD holds three waves of each, and a whole cfg3 frame gains 3.5 % with 64 Mi slots, 5.4 % with 128 Mi (profiles/r4_experiments_ab.txt); the
    // general builds (demo: 128 + 117 registers) do not fit beside each other and lose 2 %: one pool for those
    const bool lean_pair = s->leaf_level == 0 && s->ds.shade_lean != 0 && mode == 0;
    const int pools = c->st_pools > 0 ? c->st_pools : ((sharded || lean_pair) ? 2 : 1);
    hipError_t e = zr::stream_render(s->ds, dc, de, seed, c->d_pool.p, P, spp, n_pix, c->d_pixels.p, c->d_partial.p, c->d_ctl.p,
                                     c->d_st_overflow.p, c->st_ovf_levels, c->st_blocks, d_out, c->d_ctr.p, count != 0, streams, pools, c->st_event, &timer, c->h_active,
                                     keep_going, &rounds, s->leaf_level, mode, mode ? (void*)c->d_kend.p : nullptr, mode ? (void*)c->d_cls.p : nullptr, d_out2, mode ? c->d_cpart.p : nullptr, progress,
                                     use_drain ? (void*)((unsigned char*)c->d_pool.p + drain_at) : nullptr, drain_slots, unit_chunk);
    if (e != hipSuccess) return fail(ZR_E_DEVICE, "streaming pipeline failed: %s", hipGetErrorString(e));
    c->last_rounds = (uint64_t)(rounds < 0 ? -rounds : rounds);
    HIP_OK(hipStreamSynchronize(stream));
    if (rounds < 0) return fail(ZR_E_CANCELLED, "render cancelled after %d rounds", -rounds);
    return ZR_OK;
}

int enqueue_render(zr_ctx* c, const zr_scene* s, const zr_camera* cam, const zr_env* env, uint64_t seed, const Plan& plan, int count,
                   double* d_out, hipStream_t stream, volatile const uint8_t* keep_going, volatile int* rows_done, zr::StreamProgress* progress = nullptr) {
    c->last_rounds = 0;
    zr::DCamera dc; make_camera(*cam, dc);
    zr::DEnv de; make_env(*env, de);
    if (de.mode > ZR_ENV_SOLID_COLOR) return fail(ZR_E_INVALID, "unknown environment mode %u", de.mode);
    if (de.mode == ZR_ENV_HDR_MAP && de.hdr_tex != ZR_NO_TEXTURE) {
        if (de.hdr_tex >= s->textures.size()) return fail(ZR_E_INVALID, "environment texture id out of range");
    }
    std::vector<int32_t> tiles = plan.tiles;
    int rc = c->d_tiles.upload(tiles);
    if (rc) return rc;
    HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 16 * sizeof(unsigned long long), stream));
    // the streaming pipeline packs bounce counters into 8 bits, work units into 32 bits and leaf references into 24 + 4
    // bits; frames or scenes beyond that are rendered by the pixel-group megakernel below (slower, same results)
    uint64_t stream_units = 0;
    for (int32_t t : plan.tiles) {
        int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
        int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
        if (xb > xa && yb > ya) stream_units += (uint64_t)(xb - xa) * (yb - ya) * (uint64_t)dc.spp;
    }
    if (c->variant == 2 && dc.max_depth <= 250 && s->quad_ok && stream_units <= 0xFFFFFFFFull && plan.W <= 65535 && plan.H <= 65535) {
        int r2 = render_stream(c, s, dc, de, seed, plan, count, d_out, stream, keep_going, 0, nullptr, progress);
        if (rows_done && r2 == ZR_OK) *rows_done = plan.H;
        if (r2 != ZR_E_NOMEM) return r2;
        // the pipeline's buffers (24 bytes per primary sample + the slot pool) do not fit beside what else lives on the device: the
        // pixel-group kernel below needs neither
        std::fprintf(stderr, "[zr] %s: rendering this frame with the pixel-group kernel (same results, slower)\n", g_err.c_str());
        HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 16 * sizeof(unsigned long long), stream));
    } else if (c->variant == 2 && !c->warned_fallback) {   // said once per context: the frame is rendered, by the slower kernel
        c->warned_fallback = true;
        std::fprintf(stderr, "[zr] frame outside the streaming pipeline's packing limits (max_depth %d > 250, %llu work units > 2^32, %d x %d px > 65535, "
                             "or a scene with more than 2^24 primitives of a kind): rendered by the pixel-group kernel — same results, about six times slower\n",
                     dc.max_depth, (unsigned long long)stream_units, plan.W, plan.H);
    }
    c->last_path = 0;
    // one launch per frame unless the caller wants progress / cancellation, which need batch boundaries
    const bool interactive = keep_going || rows_done;
    const int batch = std::max(1, (int)env_double("ZR_BATCH_TILES", interactive ? 256 : (double)(1 << 30)));
    size_t n_batches = (tiles.size() + batch - 1) / batch;
    if (c->pending.size() > 4096) { int rr = resolve_times(c); if (rr) return rr; }
    c->render_id++; c->last_stream = stream; c->last_counted = count != 0;
    auto get_event = [&](hipEvent_t& e) -> int {
        if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); return ZR_OK; }
        HIP_OK(hipEventCreate(&e));
        return ZR_OK;
    };
    for (size_t b = 0; b < n_batches; b++) {
        if (keep_going && *keep_going == 0) {
            HIP_OK(hipStreamSynchronize(stream));
            return fail(ZR_E_CANCELLED, "render cancelled after %zu of %zu batches", b, n_batches);
        }
        zr::WorkDesc wd;
        wd.tiles = c->d_tiles.p + b * batch;
        wd.n_tiles = (int32_t)std::min<size_t>(batch, tiles.size() - b * batch);
        wd.tile_size = plan.ts; wd.tiles_x = plan.tiles_x;
        wd.x0 = plan.x0; wd.y0 = plan.y0; wd.x1 = plan.x1; wd.y1 = plan.y1;
        wd.lanes_per_pixel = plan.lanes;
        zr_ctx::Pending pe{}; pe.render_id = c->render_id; pe.kind = 1;
        if ((rc = get_event(pe.a)) || (rc = get_event(pe.b))) return rc;
        HIP_OK(hipEventRecord(pe.a, stream));
        HIP_OK(zr::launch_render(s->ds, dc, de, seed, wd, d_out, c->d_ctr.p, count != 0, stream));
        HIP_OK(hipEventRecord(pe.b, stream));
        c->pending.push_back(pe);
        if (keep_going || rows_done) {
            // progress / cancellation need the batch to have finished (camera.hpp:441,548-552)
            HIP_OK(hipStreamSynchronize(stream));
            if (rows_done) {
                int last_tile = tiles[std::min(tiles.size(), (b + 1) * (size_t)batch) - 1];
                int rows = std::min(plan.H, (last_tile / plan.tiles_x) * plan.ts);
                if (rows > *rows_done) *rows_done = rows;
            }
        }
    }
    return ZR_OK;
}

int resolve_times(zr_ctx* c) {
    if (c->pending.empty()) return ZR_OK;
    bool fresh = false;
    for (auto& p : c->pending) {
        HIP_OK(hipEventSynchronize(p.b));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, p.a, p.b));
        if (p.kind == c->log_kind) c->log.push_back(ms);  // default: the dominant kernel's launches (render_* / stream_extend)
        if (p.render_id == c->render_id) {
            if (!fresh) { c->last_render_ms = 0; c->last_extend_ms = 0; c->last_shade_ms = 0; fresh = true; }
            c->last_render_ms += ms;
            if (p.kind == 1) c->last_extend_ms += ms;
            if (p.kind == 2) c->last_shade_ms += ms;
        }
        c->pool.push_back(p.a); c->pool.push_back(p.b);
    }
    c->pending.clear();
    if (c->log.size() > (1u << 20)) c->log.erase(c->log.begin(), c->log.begin() + (c->log.size() - (1u << 20)));
    return ZR_OK;
}

}  // namespace

extern "C" {

int zr_render_device(zr_ctx* c, const zr_scene* s, const zr_camera* cam, const zr_env* env, uint64_t seed, const zr_region* region,
                     int collect_counters, void* d_out_rgb, void* hip_stream) {
    if (!c || !s || !cam || !env || !d_out_rgb) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_render");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    HIP_OK(hipSetDevice(c->device));
    Plan plan;
    int rc = make_plan(*cam, region, plan);
    if (rc) return rc;
    // default stream requested: use the legacy null stream so that callers' stream-ordered work (torch) sees it
    hipStream_t st = (hipStream_t)hip_stream;
    return enqueue_render(c, s, cam, env, seed, plan, collect_counters, (double*)d_out_rgb, st, nullptr, nullptr);
}

int zr_render(zr_ctx* c, const zr_scene* s, const zr_camera* cam, const zr_env* env, uint64_t seed, const zr_region* region,
              int collect_counters, double* out_rgb, volatile const uint8_t* keep_going, volatile int* rows_done) {
    if (!c || !s || !cam || !env || !out_rgb) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_render");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    HIP_OK(hipSetDevice(c->device));
    Plan plan;
    int rc = make_plan(*cam, region, plan);
    if (rc) return rc;
    const size_t npx = (size_t)plan.W * plan.H;
    if ((rc = c->d_out.alloc(npx * 3))) return rc;
    HIP_OK(hipMemsetAsync(c->d_out.p, 0, npx * 3 * sizeof(double), c->stream));
    if (rows_done) *rows_done = 0;
    // a whole frame goes straight into the caller's buffer; a region through a staging copy (only its pixels may be touched)
    const bool whole = (size_t)plan.tiles.size() == (size_t)plan.tiles_x * plan.tiles_y && plan.x0 == 0 && plan.y0 == 0 && plan.x1 == plan.W && plan.y1 == plan.H;
    std::vector<double> frame(whole ? 0 : npx * 3);
    auto copy_out = [&]() -> int {   // the region's pixels of the device frame -> the caller's buffer
        if (whole) { HIP_OK(hipMemcpy(out_rgb, c->d_out.p, npx * 3 * sizeof(double), hipMemcpyDeviceToHost)); return ZR_OK; }
        HIP_OK(hipMemcpy(frame.data(), c->d_out.p, frame.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int32_t t : plan.tiles) {
            int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
            int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
            for (int y = ya; y < yb; y++)
                if (xb > xa) std::memcpy(out_rgb + ((size_t)y * plan.W + xa) * 3, frame.data() + ((size_t)y * plan.W + xa) * 3, (size_t)(xb - xa) * 3 * sizeof(double));
        }
        return ZR_OK;
    };
    // Progress as the reference's callers see it: lines_rendered advances while the frame renders (camera.hpp:548-552) and the
    // GUI reads render_accumulator mid-render (main.cpp:1576).  The pipeline finishes samples all over the frame rather than
    // row by row, so `rows_done` = H x the finished fraction of the samples (H only at the very end), and a few times per second
    // out_rgb receives the mean of the samples finished so far (every pixel brightens towards its final value).
    struct Preview : zr::StreamProgress {
        volatile int* rows; int H; double last = 0; std::function<int()> copy; double period;
        static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
        bool wants_frame() override { return now() - last >= period; }
        void report(double f, bool reduced) override {
            const int r = std::min(H - 1, std::max(0, (int)(f * H)));
            if (r > *rows) *rows = r;
            if (reduced) { (void)copy(); last = now(); }
        }
    } preview;
    preview.rows = rows_done; preview.H = plan.H; preview.copy = copy_out; preview.period = env_double("ZR_PREVIEW_PERIOD_S", 0.2); preview.last = Preview::now();
    int rrc = enqueue_render(c, s, cam, env, seed, plan, collect_counters, c->d_out.p, c->stream, keep_going, rows_done, rows_done ? &preview : nullptr);
    if (rrc != ZR_OK && rrc != ZR_E_CANCELLED) return rrc;
    std::string cancel_msg = g_err;
    HIP_OK(hipStreamSynchronize(c->stream));
    if ((rc = copy_out())) return rc;
    if (rrc == ZR_E_CANCELLED) { g_err = cancel_msg; return rrc; }
    if (rows_done) *rows_done = plan.H;  // camera.hpp:576-578
    return ZR_OK;
}

int zr_render_aov(zr_ctx* c, const zr_scene* s, const zr_camera* cam, uint64_t seed, const zr_region* region, const zr_aov_params* ap,
                  double* out_albedo, double* out_normal, double* out_zdepth) {
    if (!c || !s || !cam || !ap) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_render_aov");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    if (!out_albedo && !out_normal && !out_zdepth) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    Plan plan;
    int rc = make_plan(*cam, region, plan);
    if (rc) return rc;
    zr::DCamera dc; make_camera(*cam, dc);
    // camera basis u, v, w exactly as camera::initialize builds it (camera.hpp:380-382)
    H3 w = unit(h3(cam->lookfrom) - h3(cam->lookat));
    H3 u = unit(cross(h3(cam->vup), w));
    H3 v = cross(w, u);
    double uvw[9] = {u.x, u.y, u.z, v.x, v.y, v.z, w.x, w.y, w.z};
    const int spp = dc.spp;
    const int aux_sample = std::min(std::max(spp / 8, 64), 1024);   // std::clamp(spp / 8, 64, 1024), camera.hpp:433
    const int aux = std::min(aux_sample, spp);                      // camera.hpp:535
    const size_t npx = (size_t)plan.W * plan.H;
    DevBuf<double> d_a, d_n, d_z;
    if (out_albedo) { if ((rc = d_a.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_a.p, 0, npx * 24, c->stream)); }
    if (out_normal) { if ((rc = d_n.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_n.p, 0, npx * 24, c->stream)); }
    if (out_zdepth) { if ((rc = d_z.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_z.p, 0, npx * 24, c->stream)); }
    std::vector<int32_t> tiles = plan.tiles;
    if ((rc = c->d_tiles.upload(tiles))) return rc;
    zr::WorkDesc wd;
    wd.tiles = c->d_tiles.p; wd.n_tiles = (int32_t)tiles.size(); wd.tile_size = plan.ts; wd.tiles_x = plan.tiles_x;
    wd.x0 = plan.x0; wd.y0 = plan.y0; wd.x1 = plan.x1; wd.y1 = plan.y1;
    wd.lanes_per_pixel = 64; while (wd.lanes_per_pixel > aux) wd.lanes_per_pixel >>= 1;
    HIP_OK(zr::launch_aov(s->ds, dc, seed, wd, aux, ap->z_depth_max_dist, uvw, d_a.p, d_n.p, d_z.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    std::vector<double> frame(npx * 3);
    auto copy_out = [&](DevBuf<double>& d, double* out) -> int {
        if (!out) return ZR_OK;
        HIP_OK(hipMemcpy(frame.data(), d.p, frame.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int32_t t : plan.tiles) {
            int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
            int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
            for (int y = ya; y < yb; y++)
                if (xb > xa) std::memcpy(out + ((size_t)y * plan.W + xa) * 3, frame.data() + ((size_t)y * plan.W + xa) * 3, (size_t)(xb - xa) * 3 * sizeof(double));
        }
        return ZR_OK;
    };
    if ((rc = copy_out(d_a, out_albedo)) || (rc = copy_out(d_n, out_normal)) || (rc = copy_out(d_z, out_zdepth))) return rc;
    return ZR_OK;
}

int zr_render_passes(zr_ctx* c, const zr_scene* s, const zr_camera* cam, const zr_env* env, uint64_t seed, const zr_region* region,
                     double* out_beauty, double* out_reflection, double* out_refraction) {
    if (!c || !s || !cam || !env) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_render_passes");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    if (!out_beauty && !out_reflection && !out_refraction) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    Plan plan;
    int rc = make_plan(*cam, region, plan);
    if (rc) return rc;
    zr::DCamera dc; make_camera(*cam, dc);
    zr::DEnv de; make_env(*env, de);
    if (de.mode > ZR_ENV_SOLID_COLOR) return fail(ZR_E_INVALID, "unknown environment mode %u", de.mode);
    if (de.mode == ZR_ENV_HDR_MAP && de.hdr_tex != ZR_NO_TEXTURE && de.hdr_tex >= s->textures.size()) return fail(ZR_E_INVALID, "environment texture id out of range");
    const size_t npx = (size_t)plan.W * plan.H;
    DevBuf<double> d_b, d_r, d_f;
    if (out_beauty) { if ((rc = d_b.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_b.p, 0, npx * 24, c->stream)); }
    if (out_reflection) { if ((rc = d_r.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_r.p, 0, npx * 24, c->stream)); }
    if (out_refraction) { if ((rc = d_f.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_f.p, 0, npx * 24, c->stream)); }
    std::vector<int32_t> tiles = plan.tiles;
    if ((rc = c->d_tiles.upload(tiles))) return rc;
    zr::WorkDesc wd;
    wd.tiles = c->d_tiles.p; wd.n_tiles = (int32_t)tiles.size(); wd.tile_size = plan.ts; wd.tiles_x = plan.tiles_x;
    wd.x0 = plan.x0; wd.y0 = plan.y0; wd.x1 = plan.x1; wd.y1 = plan.y1;
    wd.lanes_per_pixel = 64; while (wd.lanes_per_pixel > dc.spp) wd.lanes_per_pixel >>= 1;
    const uint64_t stream_units = (uint64_t)plan.tiles.size() * plan.ts * plan.ts * (uint64_t)dc.spp;
    const bool streaming = c->variant == 2 && s->quad_ok && 2 * dc.max_depth <= 250 && stream_units <= 0xFFFFFFFFull && plan.W <= 65535 &&
                           plan.H <= 65535 && env_double("ZR_PASSES_STREAM", 1) != 0;
    if (streaming) {
        // two runs of the streaming pipeline: the beauty pass records where every sample's stream stopped, the replay pass traces
        // the camera ray again and runs the second path from there (stream_shade MODE 1 / 2)
        unsigned long long ha[16], hb[16];
        if ((rc = render_stream(c, s, dc, de, seed, plan, 0, d_b.p, c->stream, nullptr, 1))) return rc;
        HIP_OK(hipMemcpy(ha, c->d_ctr.p, sizeof ha, hipMemcpyDeviceToHost));
        if ((rc = render_stream(c, s, dc, de, seed, plan, 0, d_r.p, c->stream, nullptr, 2, d_f.p))) return rc;
        HIP_OK(hipMemcpy(hb, c->d_ctr.p, sizeof hb, hipMemcpyDeviceToHost));
        // counted by SHADE in both passes (EXTEND runs uninstrumented): samples, segments, hits, draws
        unsigned long long h[16] = {0};
        h[0] = (unsigned long long)c->d_pixels.n * (unsigned long long)dc.spp;   // every sample of the region, once
        h[1] = ha[1] + hb[1]; h[7] = ha[7] + hb[7]; h[8] = ha[8] + hb[8];
        HIP_OK(hipMemcpy(c->d_ctr.p, h, sizeof h, hipMemcpyHostToDevice));
        c->last_counted = true;
    } else {
        c->render_id++; c->last_counted = true; c->last_rounds = 0;
        HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 16 * sizeof(unsigned long long), c->stream));
        HIP_OK(zr::launch_passes(s->ds, dc, de, seed, wd, d_b.p, d_r.p, d_f.p, c->d_ctr.p, c->stream));
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    std::vector<double> frame(npx * 3);
    auto copy_out = [&](DevBuf<double>& d, double* out) -> int {
        if (!out) return ZR_OK;
        HIP_OK(hipMemcpy(frame.data(), d.p, frame.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int32_t t : plan.tiles) {
            int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
            int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
            for (int y = ya; y < yb; y++)
                if (xb > xa) std::memcpy(out + ((size_t)y * plan.W + xa) * 3, frame.data() + ((size_t)y * plan.W + xa) * 3, (size_t)(xb - xa) * 3 * sizeof(double));
        }
        return ZR_OK;
    };
    if ((rc = copy_out(d_b, out_beauty)) || (rc = copy_out(d_r, out_reflection)) || (rc = copy_out(d_f, out_refraction))) return rc;
    return ZR_OK;
}

int zr_post_process(zr_ctx* c, const zr_post_params* pp, const double* frame, int W, int H, int is_data_pass, int apply_gamma, uint8_t* out) {
    if (!c || !pp || !frame || !out) return fail(ZR_E_INVALID, "null argument");
    if (W < 2 || H < 2 || (size_t)W * H > (1ull << 31)) return fail(ZR_E_INVALID, "frame size %d x %d not supported", W, H);
    if (pp->use_bloom && (pp->bloom_radius < 0 || pp->bloom_radius > 4096)) return fail(ZR_E_INVALID, "bloom radius out of range");
    HIP_OK(hipSetDevice(c->device));
    const size_t n = (size_t)W * H;
    DevBuf<double> d_frame, t0, t1, t2; DevBuf<uint8_t> d_out;
    int rc;
    if ((rc = d_frame.alloc(n * 3)) || (rc = d_out.alloc(n * 3))) return rc;
    const bool bloom = !is_data_pass && pp->use_bloom, sharpen = !is_data_pass && pp->use_sharpening;
    if (bloom && ((rc = t0.alloc(n * 3)) || (rc = t1.alloc(n * 3)))) return rc;
    if (sharpen && (rc = t2.alloc(n * 3))) return rc;
    HIP_OK(hipMemcpyAsync(d_frame.p, frame, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const double ev = std::pow(2.0, (double)pp->exposure);   // camera.hpp:711
    HIP_OK(zr::launch_post(d_frame.p, W, H, *pp, is_data_pass, apply_gamma, ev, t0.p, t1.p, t2.p, d_out.p, c->stream));
    HIP_OK(hipMemcpyAsync(out, d_out.p, n * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return ZR_OK;
}

int zr_analyze_frame(zr_ctx* c, const double* frame, size_t n, zr_image_stats* out) {
    if (!c || !frame || !out) return fail(ZR_E_INVALID, "null argument");
    if (n == 0 || n > (1ull << 31)) return fail(ZR_E_INVALID, "pixel count not supported");
    HIP_OK(hipSetDevice(c->device));
    const size_t blocks = (n + 255) / 256;
    DevBuf<double> d_frame, d_log; DevBuf<float> d_max; DevBuf<int> d_hist;
    int rc;
    if ((rc = d_frame.alloc(n * 3)) || (rc = d_log.alloc(blocks)) || (rc = d_max.alloc(blocks)) || (rc = d_hist.alloc(256))) return rc;
    HIP_OK(hipMemcpyAsync(d_frame.p, frame, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(zr::launch_analyze(d_frame.p, n, d_log.p, d_max.p, d_hist.p, c->stream));
    std::vector<double> plog(blocks); std::vector<float> pmax(blocks);
    HIP_OK(hipMemcpyAsync(plog.data(), d_log.p, blocks * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipMemcpyAsync(pmax.data(), d_max.p, blocks * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipMemcpyAsync(out->histogram, d_hist.p, 256 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    double total = 0.0; float mx = 0.0f;
    for (size_t b = 0; b < blocks; b++) { total += plog[b]; if (pmax[b] > mx) mx = pmax[b]; }
    out->max_luminance = mx;
    out->average_luminance = std::pow(2.0f, static_cast<float>(total / (double)n));   // color_processing.hpp:180
    return ZR_OK;
}

int zr_trace_paths(zr_ctx* c, const zr_scene* s, const zr_camera* cam, uint64_t seed, const int32_t* requests, int n, int max_segments, double* out) {
    if (!c || !s || !cam || (n > 0 && (!requests || !out))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_trace_paths");
    if (n <= 0 || max_segments <= 0) return ZR_OK;
    static_assert(ZR_PATH_RECORD == ZR_PATH_REC, "record size");
    HIP_OK(hipSetDevice(c->device));
    zr::DCamera dc; make_camera(*cam, dc);
    for (int k = 0; k < n; k++)
        if (requests[3 * k] < 0 || requests[3 * k] >= dc.W || requests[3 * k + 1] < 0 || requests[3 * k + 1] >= dc.H || requests[3 * k + 2] < 0)
            return fail(ZR_E_INVALID, "path request %d outside the frame", k);
    DevBuf<int32_t> d_req; DevBuf<double> d_out;
    std::vector<int32_t> r(requests, requests + (size_t)n * 3);
    int rc;
    if ((rc = d_req.upload(r))) return rc;
    const size_t words = (size_t)n * max_segments * ZR_PATH_RECORD;
    if ((rc = d_out.alloc(words))) return rc;
    HIP_OK(zr::launch_path_records(s->ds, dc, seed, d_req.p, n, max_segments, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out, d_out.p, words * sizeof(double), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_get_counters(zr_ctx* c, zr_counters* out) {
    if (!c || !out) return fail(ZR_E_INVALID, "null argument");
    HIP_OK(hipSetDevice(c->device));
    int rc = resolve_times(c);
    if (rc) return rc;
    std::memset(out, 0, sizeof *out);
    out->kernel_ms = c->last_render_ms;
    out->extend_ms = c->last_extend_ms; out->shade_ms = c->last_shade_ms; out->rounds = c->last_rounds; out->path = (uint64_t)c->last_path;
    unsigned long long h[16];
    HIP_OK(hipMemcpy(h, c->d_ctr.p, sizeof h, hipMemcpyDeviceToHost));
    if (h[15] != 0) return fail(ZR_E_DEVICE, "render kernel hit its iteration cap on %llu task(s): results are incomplete", h[15]);
    if (c->last_counted || env_double("ZR_RAW_COUNTERS", 0) != 0) {
        out->primary_samples = h[0]; out->segments = h[1]; out->nodes_tested = h[2]; out->spheres_tested = h[3];
        out->triangles_tested = h[4]; out->cubes_tested = h[5]; out->media_tested = h[6]; out->hits = h[7]; out->rng_draws = h[8];
        out->node_execs = h[9]; out->node_lanes = h[10]; out->leaf_execs = h[11]; out->leaf_lanes = h[12]; out->shade_execs = h[13]; out->shade_lanes = h[14];
    }
    return ZR_OK;
}

int zr_get_kernel_times(zr_ctx* c, float* ms, int cap) {
    if (!c) return fail(ZR_E_INVALID, "null argument");
    HIP_OK(hipSetDevice(c->device));
    int rc = resolve_times(c);
    if (rc) return rc;
    int total = (int)c->log.size();
    int n = std::min(total, std::max(cap, 0));
    for (int k = 0; k < n; k++) ms[k] = c->log[c->log.size() - n + k];
    c->log.clear();
    return total;
}

int zr_trace(zr_ctx* c, const zr_scene* s, const double* rays6, size_t n, double tmin, double tmax, uint64_t seed, uint64_t pixel,
             uint32_t bounce, zr_hit* out) {
    if (!c || !s || (n && (!rays6 || !out))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_trace");
    HIP_OK(hipSetDevice(c->device));
    DevBuf<double> d_rays; DevBuf<zr_hit> d_hits;
    std::vector<double> r(rays6, rays6 + n * 6);
    int rc;
    if ((rc = d_rays.upload(r))) return rc;
    if ((rc = d_hits.alloc(n))) return rc;
    // Two engines answer the same question: the pair-BVH walk of variants 0/1 and — for the render interval
    // [0.001, inf) — the EXTEND kernel of the streaming pipeline.  ZR_TRACE_ENGINE=pairs|extend picks one (tests run
    // both); by default the engine of the active render variant is used.
    const char* eng = std::getenv("ZR_TRACE_ENGINE");
    const bool can_extend = c->variant == 2 && s->quad_ok && tmin == 0.001 && tmax == HUGE_VAL && n < (1u << 30);
    if (eng && std::strcmp(eng, "extend") == 0 && !can_extend)
        return fail(ZR_E_INVALID, "ZR_TRACE_ENGINE=extend needs ZR_KERNEL=2, tmin = 0.001, tmax = inf and a scene within the 4-wide tree's limits");
    if (can_extend && !(eng && std::strcmp(eng, "pairs") == 0)) {
        DevBuf<unsigned char> pool;
        if ((rc = ensure_stack_slabs(c, s))) return rc;
        if ((rc = pool.alloc(zr::stream_pool_bytes((uint32_t)n) + 65536))) return rc;
        HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 16 * sizeof(unsigned long long), c->stream));
        HIP_OK(zr::stream_trace(s->ds, d_rays.p, (uint32_t)n, seed, pixel, bounce, d_hits.p, pool.p, c->d_ctl.p, c->d_st_overflow.p, c->st_ovf_levels, c->st_blocks,
                                c->d_ctr.p, s->leaf_level, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        unsigned int capped = 0;
        HIP_OK(hipMemcpy(&capped, c->d_ctl.p + 2, sizeof capped, hipMemcpyDeviceToHost));
        if (capped) return fail(ZR_E_DEVICE, "EXTEND hit its iteration cap on %u wave(s)", capped);
    } else {
        HIP_OK(zr::launch_trace(s->ds, d_rays.p, n, tmin, tmax, seed, pixel, bounce, d_hits.p, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    if (n) HIP_OK(hipMemcpy(out, d_hits.p, n * sizeof(zr_hit), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_kat_scatter(zr_ctx* c, const zr_scene* s, const double* rays6, const zr_hit* recs, const uint64_t* keys, const uint64_t* first_draw,
                   size_t n, zr_scatter_out* out) {
    if (!c || !s || (n && (!rays6 || !recs || !keys || !out))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_kat_scatter");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    if (n == 0) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    DevBuf<double> d_rays; DevBuf<zr_hit> d_recs; DevBuf<uint64_t> d_keys, d_first; DevBuf<zr_scatter_out> d_out;
    int rc;
    if ((rc = d_rays.upload(std::vector<double>(rays6, rays6 + n * 6))) || (rc = d_recs.upload(std::vector<zr_hit>(recs, recs + n))) ||
        (rc = d_keys.upload(std::vector<uint64_t>(keys, keys + n))) || (rc = d_out.alloc(n))) return rc;
    if (first_draw && (rc = d_first.upload(std::vector<uint64_t>(first_draw, first_draw + n)))) return rc;
    HIP_OK(zr::launch_kat_scatter(s->ds, d_rays.p, d_recs.p, d_keys.p, first_draw ? d_first.p : nullptr, n, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out, d_out.p, n * sizeof(zr_scatter_out), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_kat_texture(zr_ctx* c, const zr_scene* s, uint32_t texture_id, const double* uvp5, size_t n, double* out_rgb) {
    if (!c || !s || (n && (!uvp5 || !out_rgb))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_kat_texture");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    if (texture_id >= s->textures.size()) return fail(ZR_E_INVALID, "texture id %u out of range", texture_id);
    if (n == 0) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    DevBuf<double> d_in, d_out;
    int rc;
    if ((rc = d_in.upload(std::vector<double>(uvp5, uvp5 + n * 5))) || (rc = d_out.alloc(n * 3))) return rc;
    HIP_OK(zr::launch_kat_texture(s->ds, texture_id, d_in.p, n, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out_rgb, d_out.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_kat_background(zr_ctx* c, const zr_scene* s, const zr_env* env, const double* dirs3, size_t n, double* out_rgb) {
    if (!c || !s || !env || (n && (!dirs3 || !out_rgb))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_kat_background");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    zr::DEnv de; make_env(*env, de);
    if (de.mode > ZR_ENV_SOLID_COLOR) return fail(ZR_E_INVALID, "unknown environment mode %u", de.mode);
    if (de.mode == ZR_ENV_HDR_MAP && de.hdr_tex != ZR_NO_TEXTURE && de.hdr_tex >= s->textures.size()) return fail(ZR_E_INVALID, "environment texture id out of range");
    if (n == 0) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    DevBuf<double> d_in, d_out;
    int rc;
    if ((rc = d_in.upload(std::vector<double>(dirs3, dirs3 + n * 3))) || (rc = d_out.alloc(n * 3))) return rc;
    HIP_OK(zr::launch_kat_background(s->ds, de, d_in.p, n, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out_rgb, d_out.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_kat_camera_rays(zr_ctx* c, const zr_camera* cam, uint64_t seed, const int32_t* requests3, size_t n, double* out7) {
    if (!c || !cam || (n && (!requests3 || !out7))) return fail(ZR_E_INVALID, "null argument");
    if (n == 0) return ZR_OK;
    zr::DCamera dc; make_camera(*cam, dc);
    for (size_t k = 0; k < n; k++)
        if (requests3[3 * k] < 0 || requests3[3 * k] >= dc.W || requests3[3 * k + 1] < 0 || requests3[3 * k + 1] >= dc.H || requests3[3 * k + 2] < 0)
            return fail(ZR_E_INVALID, "camera-ray request %zu outside the frame", k);
    HIP_OK(hipSetDevice(c->device));
    DevBuf<int32_t> d_req; DevBuf<double> d_out;
    int rc;
    if ((rc = d_req.upload(std::vector<int32_t>(requests3, requests3 + n * 3))) || (rc = d_out.alloc(n * 7))) return rc;
    HIP_OK(zr::launch_kat_camera_rays(dc, seed, d_req.p, n, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out7, d_out.p, n * 7 * sizeof(double), hipMemcpyDeviceToHost));
    return ZR_OK;
}

}  // extern "C"
