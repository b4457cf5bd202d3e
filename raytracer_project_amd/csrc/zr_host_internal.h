// zr_host_internal.h — what the host-side translation units of libzr_hip.so share (round 4: zr_host.cpp was one 2 482-line file):
//   zr_host.cpp    context life cycle, error string, small C-ABI entry points
//   zr_commit.cpp  scene setters, validation, zr_scene_commit through the host builder or the device builder (zr_flatten.h: the flattener)
//   zr_render.cpp  camera frame / sky constants, the render drivers (streaming pipeline, fused kernel, pixel-group kernel), AOV / split passes, post stack,
//                  known-answer entry points
// Plain C++ (host compiler, -ffp-contract=off so that the camera frame and the sky constants are computed with exactly the reference's operation order,
// camera.hpp:358-399, 874-895,914); the kernels live in the .hip files.  There is deliberately no CPU rendering path in this library: without a HIP device
// zr_create() fails and says so.
#pragma once
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

namespace zr_host {
int fail(int code, const char* fmt, ...);          // sets the calling thread's zr_last_error() text, returns `code`
const char* last_error();
double env_double(const char* name, double dflt);  // a number from the environment (development switches), `dflt` when unset
}
using zr_host::fail;
using zr_host::env_double;

#define HIP_OK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(ZR_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

namespace zr_host {

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

}  // namespace zr_host
using namespace zr_host;

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
    // memory a commit frees off the caller's clock (staging arrays, the device builder's arena): at most one such thread is outstanding, and zr_destroy joins it —
    // a detached thread could still be inside hipFree while the process tears the HIP runtime down
    std::mutex trash_m;
    std::thread trash;
    template <class F> void free_later(F&& f) {
        std::lock_guard<std::mutex> lk(trash_m);
        if (trash.joinable()) trash.join();
        try { trash = std::thread(std::forward<F>(f)); } catch (...) { f(); }
    }
    void drain_trash() { std::lock_guard<std::mutex> lk(trash_m); if (trash.joinable()) trash.join(); }
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
    int device = 0;               // the context's device ordinal, kept here so that zr_scene_destroy needs no live context (a scene may outlive its context)
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
    unsigned reset_mask = 0;      // ... which geometry arrays have been given again since (SET_* bits, zr_commit.cpp: GEOMETRY_SET)
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

struct CommitSummary {   // what the shared end of a commit needs to know about the tree either builder produced
    zr::NodeF root{}; bool quant_ok = true;
    size_t n_pairs = 0, n_quads = 0, n_sph = 0, n_tri = 0, n_cube = 0, n_pcube = 0, n_media = 0, n_wrapped = 0, n_insts = 0, kept_closed = 0;
    bool plain_media = true;
    uint32_t stack_demand = 0; int quad_depth = 0, max_depth = 0, max_leaf = 4;
    uint32_t leaf_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // leaf objects per kind
    const char* builder = "";
};
// the tables every scene has (ops, materials, textures), the DScene the kernels receive, the EXTEND build the world needs
