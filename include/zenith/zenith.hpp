// zenith.hpp — the drop-in C++20 scene API of the MI355X integrator.
//
// Same class names, constructors and public signatures as the reference's scene-authoring and render
// API (global namespace, like the reference), so that code written against
//   /root/reference/{vec3,ray,interval,aabb,hittable,hittable_list,sphere,triangle,cube,constant_medium,
//                     translate,rotate_x,rotate_y,rotate_z,scale,material_instance,bvh,texture,material,
//                     environment,camera}.hpp
// compiles unchanged (scenes/zr_scenes.inc is compiled against both).  What differs, by design:
//   * objects EXPOSE their data through flatten(zenith::scene_builder&) — reference objects keep all
//     members private with no getters (e.g. sphere.hpp:85-89), so a device backend cannot introspect them;
//   * camera::render() does not run a CPU loop: it flattens the world into the arrays of
//     include/zr_capi.h and calls the C ABI (libzr_hip.so); if that library or a HIP device is
//     missing it reports the error on std::cerr and leaves the buffers zeroed — there is NO CPU fallback;
//   * hittable::hit() / material::scatter() of the built-in classes stay CALLABLE, but they are not CPU-evaluated: a
//     call flattens the object (cached in the object), commits it as a one-object device scene and sends the ray
//     through the same kernels the renderer uses (zr_trace / zr_kat_scatter of the C ABI).  One launch per call — the
//     per-ray virtual path is exactly what this drop-in replaces, so this is for probing and tests, never for rendering;
//   * random_double() draws from the seedable contract stream of include/zr_rng.h instead of a racy
//     process-global mt19937 (common.hpp:29-34).
// No code is taken from the reference; each class cites the interface it mirrors.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <typeinfo>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../zr_capi.h"
#include "../zr_rng.h"
#ifdef ZENITH_STB_IMAGE
#include "stb_image.h"
#endif

using std::make_shared;
using std::shared_ptr;

// ---- common.hpp:14-44 ---------------------------------------------------------------------------
constexpr double infinity = std::numeric_limits<double>::infinity();
constexpr double pi = 3.14159265358979323846;
const double ray_epsilon = 0.0001;
inline double degrees_to_radians(double d) { return d * pi / 180.0; }
inline double radians_to_degrees(double r) { return r * 180.0 / pi; }

namespace zenith {
struct host_rng { uint64_t key = zr_stream_key(0, ZR_SCENE_PIXEL, 0); uint64_t k = 0; };
inline host_rng& rng_state() { static thread_local host_rng s; return s; }
// extension (the reference has no seed API): position random_double() on a contract stream
inline void seed_rng(uint64_t seed, uint64_t pixel = ZR_SCENE_PIXEL, uint64_t sample = 0) {
    rng_state().key = zr_stream_key(seed, pixel, sample); rng_state().k = 0;
}
}  // namespace zenith
inline double random_double() { auto& s = zenith::rng_state(); return zr_bits_to_unit(zr_stream_bits(s.key, s.k++)); }
namespace zenith {
// The reference's bvh_node constructor draws random_int(0, 2) once per node it creates (bvh.hpp:17): one node per span of 1 or 2
// objects, 1 + left + right for longer spans split at span / 2 (bvh.hpp:25-41).  The drop-in builds its own tree at commit, but
// whoever constructs a bvh_node or loads a model (model.hpp:95) must find random_double() where the reference leaves it, or
// every random scene decision made afterwards differs: the constructors below consume exactly that many draws.
inline uint64_t bvh_ctor_draws(size_t n) { return n <= 2 ? (n ? 1 : 0) : 1 + bvh_ctor_draws(n / 2) + bvh_ctor_draws(n - n / 2); }
inline void consume_draws(uint64_t n) { rng_state().k += n; }
}  // namespace zenith
inline double random_double(double lo, double hi) { return lo + (hi - lo) * random_double(); }
inline int random_int(int lo, int hi) { return static_cast<int>(random_double(lo, hi + 1)); }

// ---- vec3.hpp ------------------------------------------------------------------------------------
class vec3 {
public:
    double e[3];
    constexpr vec3() : e{0, 0, 0} {}
    constexpr vec3(double a, double b, double c) : e{a, b, c} {}
    constexpr double x() const { return e[0]; }
    constexpr double y() const { return e[1]; }
    constexpr double z() const { return e[2]; }
    constexpr vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    constexpr double operator[](int i) const { return e[i]; }
    double& operator[](int i) { return e[i]; }
    vec3& operator+=(const vec3& o) { e[0] += o.e[0]; e[1] += o.e[1]; e[2] += o.e[2]; return *this; }
    vec3& operator-=(const vec3& o) { e[0] -= o.e[0]; e[1] -= o.e[1]; e[2] -= o.e[2]; return *this; }
    vec3& operator*=(const vec3& o) { e[0] *= o.e[0]; e[1] *= o.e[1]; e[2] *= o.e[2]; return *this; }
    vec3& operator*=(double t) { e[0] *= t; e[1] *= t; e[2] *= t; return *this; }
    vec3& operator/=(double t) { return *this *= 1 / t; }
    constexpr double length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    double length() const { return std::sqrt(length_squared()); }
    bool near_zero() const { return std::fabs(e[0]) < 1e-8 && std::fabs(e[1]) < 1e-8 && std::fabs(e[2]) < 1e-8; }
    double luminance() const { return 0.2126 * e[0] + 0.7152 * e[1] + 0.0722 * e[2]; }
    static vec3 random() { double a = random_double(), b = random_double(), c = random_double(); return vec3(a, b, c); }
    static vec3 random(double lo, double hi) {
        double a = random_double(lo, hi), b = random_double(lo, hi), c = random_double(lo, hi);
        return vec3(a, b, c);
    }
};
using point3 = vec3;
using color = vec3;
inline std::ostream& operator<<(std::ostream& o, const vec3& v) { return o << v.e[0] << " " << v.e[1] << " " << v.e[2]; }
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline vec3 operator*(double t, const vec3& v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3& v, double t) { return t * v; }
inline vec3 operator/(const vec3& v, double t) { return (1 / t) * v; }
inline double dot(const vec3& a, const vec3& b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
inline vec3 cross(const vec3& a, const vec3& b) {
    return vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
inline vec3 unit_vector(const vec3& v) { double l = v.length(); return l < 1e-8 ? vec3(0, 0, 0) : v / l; }
inline vec3 random_in_unit_disk() {
    for (;;) { double a = random_double(-1, 1), b = random_double(-1, 1); vec3 p(a, b, 0); if (p.length_squared() < 1) return p; }
}
inline vec3 random_unit_vector() {
    for (;;) { vec3 p = vec3::random(-1, 1); double l2 = p.length_squared(); if (1e-160 < l2 && l2 <= 1) return p / std::sqrt(l2); }
}
inline vec3 random_on_hemisphere(const vec3& n) { vec3 s = random_unit_vector(); return dot(s, n) > 0.0 ? s : -s; }
inline vec3 reflect(const vec3& v, const vec3& n) { return v - 2 * dot(v, n) * n; }
inline vec3 refract(const vec3& uv, const vec3& n, double eta) {
    double ct = std::fmin(dot(-uv, n), 1.0);
    vec3 perp = eta * (uv + ct * n);
    return perp + (-std::sqrt(std::fabs(1.0 - perp.length_squared())) * n);
}

// ---- ray.hpp, interval.hpp, aabb.hpp --------------------------------------------------------------
class ray {
public:
    point3 orig; vec3 dir; double tm = 0.0;
    ray() {}
    ray(const point3& o, const vec3& d) : orig(o), dir(d), tm(0.0) {}
    ray(const point3& o, const vec3& d, double t) : orig(o), dir(d), tm(t) {}
    const point3& origin() const { return orig; }
    const vec3& direction() const { return dir; }
    double time() const { return tm; }
    point3 at(double t) const { return orig + t * dir; }
};

class interval {
public:
    double min, max;
    interval() : min(+infinity), max(-infinity) {}
    interval(double lo, double hi) : min(lo), max(hi) {}
    interval(const interval& a, const interval& b) : min(std::fmin(a.min, b.min)), max(std::fmax(a.max, b.max)) {}
    interval expand(double delta) const { return interval(min - delta / 2, max + delta / 2); }
    double size() const { return max - min; }
    bool contains(double x) const { return min <= x && x <= max; }
    bool surrounds(double x) const { return min < x && max > x; }
    double clamp(double x) const { return x < min ? min : (x > max ? max : x); }
    static const interval empty, universe;
};
inline const interval interval::empty = interval(+infinity, -infinity);
inline const interval interval::universe = interval(-infinity, +infinity);
inline interval operator+(const interval& i, double d) { return interval(i.min + d, i.max + d); }
inline interval operator+(double d, const interval& i) { return i + d; }

class aabb {
public:
    interval x, y, z;
    aabb() {}
    aabb(const interval& a, const interval& b, const interval& c) : x(a), y(b), z(c) {}
    aabb(const point3& a, const point3& b)
        : x(std::fmin(a[0], b[0]), std::fmax(a[0], b[0])), y(std::fmin(a[1], b[1]), std::fmax(a[1], b[1])),
          z(std::fmin(a[2], b[2]), std::fmax(a[2], b[2])) {}
    aabb(const aabb& a, const aabb& b) : x(a.x, b.x), y(a.y, b.y), z(a.z, b.z) {}
    const interval& axis(int n) const { return n == 1 ? y : (n == 2 ? z : x); }
};
inline aabb operator+(const aabb& b, const vec3& o) { return aabb(b.x + o.x(), b.y + o.y(), b.z + o.z()); }
inline aabb operator+(const vec3& o, const aabb& b) { return b + o; }

class material;
class texture;
class hittable;

// ---- flattening: the part the reference does not have ---------------------------------------------
namespace zenith {

constexpr uint32_t no_material = 0xFFFFFFFFu;

// vectors of the large arrays: resize() leaves new elements uninitialised (no 160 MB zero-fill by one thread before the threads of
// a bulk flatten write them; their pages are first touched by those threads)
template <class T>
struct noinit_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = noinit_allocator<U>; };
    noinit_allocator() = default;
    template <class U> noinit_allocator(const noinit_allocator<U>&) {}
    template <class U, class... A> void construct(U* p, A&&... a) { if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using nvec = std::vector<T, noinit_allocator<T>>;

struct flat_scene {
    std::vector<double> spheres, cubes;
    nvec<double> tri_v, tri_n;
    std::vector<uint32_t> sphere_mat, cube_mat;
    nvec<uint32_t> tri_mat;
    std::vector<zr_medium> media;
    std::vector<zr_xform_op> ops;
    nvec<zr_object> objects;
    std::vector<zr_group> groups;
    std::vector<zr_material> materials;
    std::vector<zr_texture> textures;
    std::vector<unsigned char> texels;
    std::vector<std::string> warnings;
    // empties the scene but keeps the arrays' memory: a camera flattens its world on every render, and a quarter of a gigabyte of
    // freshly mapped pages costs more to touch than a million triangles cost to copy
    void clear() {
        spheres.clear(); cubes.clear(); tri_v.clear(); tri_n.clear(); sphere_mat.clear(); cube_mat.clear(); tri_mat.clear(); media.clear(); ops.clear();
        objects.clear(); groups.clear(); materials.clear(); textures.clear(); texels.clear(); warnings.clear();
    }
    zr_scene_desc desc() const {
        zr_scene_desc d{};
        d.spheres = spheres.data(); d.sphere_mat = sphere_mat.data(); d.n_spheres = sphere_mat.size();
        d.tri_v = tri_v.data(); d.tri_n = tri_n.data(); d.tri_mat = tri_mat.data(); d.n_tris = tri_mat.size();
        d.cubes = cubes.data(); d.cube_mat = cube_mat.data(); d.n_cubes = cube_mat.size();
        d.media = media.data(); d.n_media = media.size();
        d.ops = ops.data(); d.n_ops = ops.size();
        d.objects = objects.data(); d.n_objects = objects.size();
        d.materials = materials.data(); d.n_materials = materials.size();
        d.textures = textures.data(); d.n_textures = textures.size();
        d.texels = texels.data(); d.texel_bytes = texels.size();
        d.groups = groups.data(); d.n_groups = groups.size();
        return d;
    }
};

class scene_builder {
public:
    explicit scene_builder(flat_scene& f) : fs(f) {}
    flat_scene& fs;

    void push_op(uint32_t kind, double a0, double a1, double a2, uint32_t mat = 0) {
        zr_xform_op op{}; op.kind = kind; op.mat = mat; op.a[0] = a0; op.a[1] = a1; op.a[2] = a2; chain.push_back(op);
    }
    void pop_op() { chain.pop_back(); }

    uint32_t texture_id(const shared_ptr<texture>& t);
    uint32_t material_id(const shared_ptr<material>& m);
    uint32_t add_material(const zr_material& m) { fs.materials.push_back(m); return (uint32_t)fs.materials.size() - 1; }
    uint32_t add_texture(const zr_texture& t) { fs.textures.push_back(t); return (uint32_t)fs.textures.size() - 1; }
    uint64_t add_texels(const void* p, size_t bytes) {
        size_t off = (fs.texels.size() + 15) & ~size_t(15);
        fs.texels.resize(off + bytes);
        std::memcpy(fs.texels.data() + off, p, bytes);
        return off;
    }

    void emit_sphere(const point3& c, double radius_arg, const shared_ptr<material>& m) {
        fs.spheres.insert(fs.spheres.end(), {c.x(), c.y(), c.z(), radius_arg});
        fs.sphere_mat.push_back(material_id(m));
        emit(ZR_PRIM_SPHERE, (uint32_t)fs.sphere_mat.size() - 1);
    }
    // (the hot call of a flatten: a million-triangle world comes through here once per render, on the caller's clock)
    void emit_triangle(const point3 v[3], const vec3 n[3], const shared_ptr<material>& m) {
        const size_t i = fs.tri_mat.size();
        fs.tri_v.resize(i * 9 + 9); fs.tri_n.resize(i * 9 + 9);
        double* tv = fs.tri_v.data() + i * 9; double* tn = fs.tri_n.data() + i * 9;
        for (int k = 0; k < 3; k++) { tv[3 * k] = v[k].x(); tv[3 * k + 1] = v[k].y(); tv[3 * k + 2] = v[k].z(); tn[3 * k] = n[k].x(); tn[3 * k + 1] = n[k].y(); tn[3 * k + 2] = n[k].z(); }
        fs.tri_mat.push_back(material_id(m));
        emit(ZR_PRIM_TRIANGLE, (uint32_t)i);
    }
    // BULK: n bare triangles of one list in one go (hittable_list::flatten on long runs).  Returns false when the builder is in a
    // state where entries are not plain world entries (capturing a template, inside a medium's boundary): the caller then emits
    // them one by one.  Otherwise the arrays are grown by n, `first_tri` / `first_obj` say where, and every world entry of the run
    // shares ONE copy of the open wrapper chain.
    bool begin_bulk_triangles(size_t n, size_t& first_tri, size_t& first_obj, uint32_t& chain_first, uint32_t& chain_count, bool& entries) {
        if (in_boundary) return false;
        if (!caps.empty() && chain.size() != caps.back().base) return false;   // inside a template under wrappers of its own: members, one by one
        entries = caps.empty();   // inside a template the triangles are a run of the template, not world entries
        first_tri = fs.tri_mat.size(); first_obj = fs.objects.size();
        fs.tri_v.resize((first_tri + n) * 9); fs.tri_n.resize((first_tri + n) * 9); fs.tri_mat.resize(first_tri + n);
        chain_count = 0; chain_first = 0;
        if (entries) { fs.objects.resize(first_obj + n); chain_count = (uint32_t)chain.size(); chain_first = copy_chain(0, chain.size()); }
        return true;
    }
    void end_bulk_triangles(size_t first_tri, size_t run, size_t n_obj, bool entries) {   // the run is `run` long: cut the arrays back; in a template, note it
        fs.tri_v.resize((first_tri + run) * 9); fs.tri_n.resize((first_tri + run) * 9); fs.tri_mat.resize(first_tri + run);
        if (entries) { fs.objects.resize(n_obj); return; }
        tmpl& t = templates[caps.back().tmpl];
        if (!t.items.empty() && t.items.back().what == 0 && t.items.back().inner.empty() && t.items.back().index + t.items.back().count == first_tri) { t.items.back().count += (uint32_t)run; return; }
        tmpl_item m; m.what = 0; m.type = ZR_PRIM_TRIANGLE; m.index = (uint32_t)first_tri; m.count = (uint32_t)run;
        t.items.push_back(std::move(m));
    }
    // a list of n objects is about to be flattened: make room once instead of growing by doubling (mostly triangles in the lists
    // that are long enough to matter; capacity that stays unused is never touched)
    void reserve_hint(size_t n) {
        if (n < 4096) return;
        fs.tri_v.reserve(fs.tri_v.size() + n * 9); fs.tri_n.reserve(fs.tri_n.size() + n * 9); fs.tri_mat.reserve(fs.tri_mat.size() + n);
        if (caps.empty()) fs.objects.reserve(fs.objects.size() + n);
    }
    void emit_cube(const vec3& he, const point3& c, const point3& mn, const point3& mx, const shared_ptr<material>& m) {
        fs.cubes.insert(fs.cubes.end(), {he.x(), he.y(), he.z(), c.x(), c.y(), c.z(), mn.x(), mn.y(), mn.z(), mx.x(), mx.y(), mx.z()});
        fs.cube_mat.push_back(material_id(m));
        if (!in_boundary && !(c.x() == 0 && c.y() == 0 && c.z() == 0))
            fs.warnings.push_back("cube not centred at the origin: the reference tests it as if it were (cube.hpp:45-58) but culls "
                                  "with its true box; only origin-centred cubes (wrapped in translate) are reproduced");
        emit(ZR_PRIM_CUBE, (uint32_t)fs.cube_mat.size() - 1);
    }
    void emit_medium(const hittable& boundary, double density, uint32_t iso_material);

    void unsupported(const char* what) { fs.warnings.push_back(std::string("unsupported hittable skipped: ") + what); }

    // TWO-LEVEL BVH.  What the reference shares between instances — the same shared_ptr (a model, a bvh_node, a prefab list) inside
    // several translate / rotate_* / scale / material_instance objects — is flattened ONCE, in its own space, into a TEMPLATE:
    //   * its runs of bare triangles become zr_groups (stored once, a tree of their own on the device; every placement of the run
    //     is ONE ZR_PRIM_GROUP world entry carrying the wrapper chain),
    //   * its other members — spheres, cubes, media, primitives under wrappers of their own inside the child — are stored once
    //     too and referenced by index: a placement emits one world entry per member whose chain is the placement's followed by
    //     the member's inner chain,
    //   * a shared child INSIDE the child (a group inside a group) is a template of its own, referenced under its inner chain:
    //     its triangle runs are the same zr_groups wherever the outer child is placed.
    // So a composite prefab placed N times costs N small entries per member and per mesh, and its triangles once (round 2 shared
    // only children that were nothing but bare triangles; anything else was copied per placement, triangles included).
    // `children` flattens the child's parts.  A run placed only once in the end is dissolved by finish().
    template <class F>
    void emit_run(const void* identity, F&& children) {
        const char* e_ = std::getenv("ZR_GROUPS");
        const bool off = e_ && *e_ == '0';   // ZR_GROUPS=0: always one entry per primitive
        if (off || in_boundary || (caps.empty() && chain.empty())) { children(); return; }   // (a child added bare to the world is flattened in place)
        auto it = tmpl_ids.find(identity);
        if (it == tmpl_ids.end()) {
            const uint32_t id = (uint32_t)templates.size();
            templates.emplace_back();
            caps.push_back(capture{id, chain.size()});
            children();
            caps.pop_back();
            close_template(id);
            it = tmpl_ids.emplace(identity, id).first;
        }
        if (!caps.empty()) {   // met while an outer child is being captured: a member of that template, under the chain opened since
            tmpl_item m; m.what = 2; m.index = it->second;
            m.inner.assign(chain.begin() + (std::ptrdiff_t)caps.back().base, chain.end());
            templates[caps.back().tmpl].items.push_back(std::move(m));
            return;
        }
        expand(it->second);
    }
    // call once, after the world has been flattened and before flat_scene::desc(): groups placed a single time gain nothing from
    // a tree of their own — their placement is replaced by its triangles (each under the placement's chain) unless
    // ZR_GROUP_ALWAYS is set
    void finish() {
        if (fs.groups.empty()) return;
        const char* always = std::getenv("ZR_GROUP_ALWAYS");
        std::vector<uint32_t> uses(fs.groups.size(), 0);
        for (const zr_object& o : fs.objects) if (o.type == ZR_PRIM_GROUP) uses[o.index]++;
        std::vector<uint32_t> new_id(fs.groups.size(), 0xFFFFFFFFu);
        std::vector<zr_group> kept;
        for (size_t g = 0; g < fs.groups.size(); g++)
            if (uses[g] > 1 || (uses[g] == 1 && always && *always && *always != '0')) { new_id[g] = (uint32_t)kept.size(); kept.push_back(fs.groups[g]); }
        nvec<zr_object> out;
        out.reserve(fs.objects.size());
        for (const zr_object& o : fs.objects) {
            if (o.type != ZR_PRIM_GROUP) { out.push_back(o); continue; }
            if (new_id[o.index] != 0xFFFFFFFFu) { zr_object q = o; q.index = new_id[o.index]; out.push_back(q); continue; }
            const zr_group& g = fs.groups[o.index];
            for (uint32_t k = 0; k < g.triangle_count; k++) { zr_object q = o; q.type = ZR_PRIM_TRIANGLE; q.index = g.first_triangle + k; out.push_back(q); }
        }
        fs.objects.swap(out);
        fs.groups.swap(kept);
    }

private:
    std::vector<zr_xform_op> chain;  // wrappers currently open, outermost first
    size_t boundary_base = 0;
    bool in_boundary = false;
    zr_object captured{};            // the boundary primitive of the medium being flattened
    bool captured_ok = false;
    // templates of shared children (emit_run)
    struct tmpl_item { int what = 1; /* 0 run of triangles -> group, 1 primitive, 2 another template */ uint32_t type = 0, index = 0, count = 0; std::vector<zr_xform_op> inner; };
    struct tmpl { std::vector<tmpl_item> items; };
    struct capture { uint32_t tmpl; size_t base; /* chain.size() when the capture began: what follows is the member's inner chain */ };
    std::vector<tmpl> templates;
    std::vector<capture> caps;
    std::unordered_map<const void*, uint32_t> tmpl_ids;   // shared child -> its template
    std::unordered_map<const material*, uint32_t> mat_ids;
    const material* last_mat_ = nullptr; uint32_t last_mat_id_ = 0;   // the last lookup of material_id()
    std::unordered_map<const texture*, uint32_t> tex_ids;
public:
    std::unordered_map<uint32_t, shared_ptr<material>> mat_ptrs;   // flattened material id -> the object it came from
private:
    // a captured child is complete: its triangle runs become groups (a run shorter than ZR_GROUP_MIN_TRIS, default 4, beside other
    // members is not worth a tree of its own: it stays a list of triangles)
    void close_template(uint32_t id) {
        tmpl& t = templates[id];
        const char* e = std::getenv("ZR_GROUP_MIN_TRIS");
        const uint32_t min_tris = e && *e ? (uint32_t)std::atoi(e) : 4u;
        const bool pure = t.items.size() == 1 && t.items[0].what == 0;   // nothing but one run of triangles (a model): always a group
        std::vector<tmpl_item> out;
        for (tmpl_item& m : t.items) {
            if (m.what != 0) { out.push_back(std::move(m)); continue; }
            if (pure || m.count >= min_tris) {
                zr_group g{}; g.first_triangle = m.index; g.triangle_count = m.count;
                fs.groups.push_back(g);
                m.index = (uint32_t)fs.groups.size() - 1;
                out.push_back(std::move(m));
            } else
                for (uint32_t k = 0; k < m.count; k++) { tmpl_item q; q.what = 1; q.type = ZR_PRIM_TRIANGLE; q.index = m.index + k; out.push_back(std::move(q)); }
        }
        t.items.swap(out);
    }
    // one placement of a template under the chain that is open now
    void expand(uint32_t id) {
        for (size_t k = 0; k < templates[id].items.size(); k++) {
            const tmpl_item m = templates[id].items[k];   // (a copy: emit() below may grow `templates`' neighbours, never this vector, but stay safe)
            for (const zr_xform_op& op : m.inner) chain.push_back(op);
            if (m.what == 0) emit(ZR_PRIM_GROUP, m.index);
            else if (m.what == 1) emit(m.type, m.index);
            else expand(m.index);
            chain.resize(chain.size() - m.inner.size());
        }
    }

    uint32_t copy_chain(size_t from, size_t to) {
        uint32_t first = (uint32_t)fs.ops.size();
        for (size_t k = from; k < to; k++) fs.ops.push_back(chain[k]);
        return first;
    }
    void emit(uint32_t type, uint32_t index) {
        zr_object o{};
        o.type = type; o.index = index;
        if (!caps.empty() && !in_boundary) {   // inside a shared child that is being captured: a member of its template, not a world entry
            tmpl& t = templates[caps.back().tmpl];
            const size_t base = caps.back().base;
            if (type == ZR_PRIM_TRIANGLE && chain.size() == base && !t.items.empty() && t.items.back().what == 0 &&
                t.items.back().index + t.items.back().count == index) { t.items.back().count++; return; }   // the run goes on
            tmpl_item m;
            m.what = (type == ZR_PRIM_TRIANGLE && chain.size() == base) ? 0 : 1;
            m.type = type; m.index = index; m.count = 1;
            m.inner.assign(chain.begin() + (std::ptrdiff_t)base, chain.end());
            t.items.push_back(std::move(m));
            return;
        }
        if (in_boundary) {
            o.chain_count = (uint32_t)(chain.size() - boundary_base);
            o.chain_first = copy_chain(boundary_base, chain.size());
            captured = o; captured_ok = true;
        } else {
            o.chain_count = (uint32_t)chain.size();
            o.chain_first = copy_chain(0, chain.size());
            fs.objects.push_back(o);
        }
    }
};

[[noreturn]] inline void no_cpu_path(const char* what) {
    throw std::logic_error(std::string(what) + ": the MI355X drop-in does not evaluate rays on the CPU; render through camera::render()");
}

// hittable::hit / material::scatter of the built-in classes, answered by the device (defined at the end of this header)
struct device_object;
}  // namespace zenith
class hit_record;
namespace zenith {
bool device_hit(const hittable& self, const ray& r, const interval& ray_t, hit_record& rec);
bool device_scatter(const material& self, const ray& r_in, const hit_record& rec, vec3& attenuation, ray& scattered);
}  // namespace zenith

// ---- hittable.hpp ---------------------------------------------------------------------------------
class hit_record {
public:
    point3 p; vec3 normal, tangent, bitangent;
    shared_ptr<material> mat;
    bool front_face = false;
    double t = 0.0, u = 0.0, v = 0.0;
    void set_face_normal(const ray& r, const vec3& outward) {
        front_face = dot(r.direction(), outward) < 0;
        normal = front_face ? outward : -outward;
    }
};

class hittable {
public:
    virtual ~hittable() = default;
    virtual bool hit(const ray& r, interval ray_t, hit_record& rec, int depth = 0, bool debug_wire = false) const = 0;
    virtual aabb bounding_box() const = 0;
    // drop-in extension: hand the object's data to the device backend.  User-defined hittables that do
    // not override it are reported and skipped.
    virtual void flatten(zenith::scene_builder& b) const { b.unsupported(typeid(*this).name()); }
    mutable shared_ptr<zenith::device_object> zr_device_cache_;   // drop-in: the committed one-object scene hit() sends rays to
};

// ---- texture.hpp ----------------------------------------------------------------------------------
class texture {
public:
    virtual ~texture() = default;
    virtual color value(double u, double v, const point3& p) const = 0;
    virtual uint32_t flatten(zenith::scene_builder& b) const = 0;
};

class solid_color : public texture {
public:
    solid_color(const color& c) : albedo(c) {}
    solid_color(double r, double g, double b) : albedo(r, g, b) {}
    color value(double, double, const point3&) const override { return albedo; }
    uint32_t flatten(zenith::scene_builder& b) const override {
        zr_texture t{}; t.kind = ZR_TEX_SOLID; t.color[0] = albedo.x(); t.color[1] = albedo.y(); t.color[2] = albedo.z();
        return b.add_texture(t);
    }
private:
    color albedo;
};

class checker_texture : public texture {
public:
    checker_texture(double scale, shared_ptr<texture> odd, shared_ptr<texture> even) : inv_scale(1.0 / scale), odd(odd), even(even) {}
    checker_texture(double scale, color c1, color c2)
        : inv_scale(1.0 / scale), odd(make_shared<solid_color>(c1)), even(make_shared<solid_color>(c2)) {}
    color value(double u, double v, const point3& p) const override {
        int s = static_cast<int>(std::floor(inv_scale * p.x())) + static_cast<int>(std::floor(inv_scale * p.y())) +
                static_cast<int>(std::floor(inv_scale * p.z()));
        return s % 2 == 0 ? even->value(u, v, p) : odd->value(u, v, p);
    }
    uint32_t flatten(zenith::scene_builder& b) const override {
        zr_texture t{}; t.kind = ZR_TEX_CHECKER; t.inv_scale = inv_scale;
        t.odd = b.texture_id(odd); t.even = b.texture_id(even);
        return b.add_texture(t);
    }
private:
    double inv_scale; shared_ptr<texture> odd, even;
};

// image_texture(filename, is_hdr): texture.hpp:14-39.  The reference decodes through stb_image; the
// drop-in has its own readers for the two formats its scenes and tests use: Radiance .hdr (RGBE, flat or
// new-style RLE scanlines) for is_hdr, binary PPM (P6, maxval 255) otherwise.  Anything else behaves
// like the reference's failed load: width = height = 0 and value() returns cyan (texture.hpp:52-54).
// A host that has stb_image on its include path (the reference vendors it under libs/stb) defines ZENITH_STB_IMAGE before
// including this header (and STB_IMAGE_IMPLEMENTATION in one translation unit, as the reference's stb_impl.cpp does): every format
// stb decodes (JPEG, PNG, ...) then loads exactly as in the reference — stbi_loadf / stbi_load with 3 channels forced
// (texture.hpp:23,31) — and the built-in readers serve as the fallback.
class image_texture : public texture {
public:
    image_texture(const char* filename, bool is_hdr = false) : hdr(is_hdr) {
        bool ok = false;
#ifdef ZENITH_STB_IMAGE
        ok = load_stb(filename);
#endif
        if (!ok) ok = is_hdr ? load_hdr(filename) : load_ppm(filename);
        if (!ok) {
            std::cerr << (is_hdr ? "ERROR: Could not load HDR: " : "ERROR: Could not load texture: ") << filename << "\n";
            width = height = 0; f32.clear(); u8.clear();
        }
    }
    color value(double u, double v, const point3&) const override {
        if (width == 0 || height == 0) return color(0.0, 1.0, 1.0);
        u = u - std::floor(u);
        int i = std::clamp(static_cast<int>(u * width), 0, width - 1);
        int j = std::clamp(static_cast<int>(v * height), 0, height - 1);
        size_t o = ((size_t)j * width + i) * 3;
        if (hdr) return color(f32[o], f32[o + 1], f32[o + 2]);
        const double s = 1.0 / 255.0;
        return color(s * u8[o], s * u8[o + 1], s * u8[o + 2]);
    }
    uint32_t flatten(zenith::scene_builder& b) const override {
        zr_texture t{}; t.kind = hdr ? ZR_TEX_IMAGE_F32 : ZR_TEX_IMAGE_U8; t.width = (uint32_t)width; t.height = (uint32_t)height;
        if (width && height) t.texel_offset = hdr ? b.add_texels(f32.data(), f32.size() * 4) : b.add_texels(u8.data(), u8.size());
        return b.add_texture(t);
    }
    int image_width() const { return width; }
    int image_height() const { return height; }
private:
    bool hdr; int width = 0, height = 0;
    std::vector<float> f32; std::vector<unsigned char> u8;

#ifdef ZENITH_STB_IMAGE
    bool load_stb(const char* fn) {   // texture.hpp:20-36
        int w = 0, h = 0, ch = 0;
        if (hdr) {
            float* d = stbi_loadf(fn, &w, &h, &ch, 3);
            if (!d) return false;
            f32.assign(d, d + (size_t)w * h * 3); stbi_image_free(d);
        } else {
            unsigned char* d = stbi_load(fn, &w, &h, &ch, 3);
            if (!d) return false;
            u8.assign(d, d + (size_t)w * h * 3); stbi_image_free(d);
        }
        width = w; height = h; return w > 0 && h > 0;
    }
#endif
    bool load_ppm(const char* fn) {
        std::ifstream f(fn, std::ios::binary);
        if (!f) return false;
        std::string magic; int w = 0, h = 0, mx = 0;
        auto token = [&](std::string& s) {
            s.clear(); int c;
            for (;;) { c = f.get(); if (c == '#') { while (c != '\n' && c != EOF) c = f.get(); } else if (!isspace(c)) break; if (c == EOF) return; }
            while (c != EOF && !isspace(c)) { s.push_back((char)c); c = f.get(); }
        };
        std::string tw, th, tm;
        token(magic); token(tw); token(th); token(tm);
        if (magic != "P6") return false;
        w = std::atoi(tw.c_str()); h = std::atoi(th.c_str()); mx = std::atoi(tm.c_str());
        if (w <= 0 || h <= 0 || mx != 255) return false;
        u8.resize((size_t)w * h * 3);
        f.read((char*)u8.data(), (std::streamsize)u8.size());
        if ((size_t)f.gcount() != u8.size()) return false;
        width = w; height = h; return true;
    }
    bool load_hdr(const char* fn) {
        std::ifstream f(fn, std::ios::binary);
        if (!f) return false;
        std::string line;
        if (!std::getline(f, line) || (line.rfind("#?RADIANCE", 0) != 0 && line.rfind("#?RGBE", 0) != 0)) return false;
        bool fmt = false;
        while (std::getline(f, line)) { if (line.empty()) break; if (line == "FORMAT=32-bit_rle_rgbe") fmt = true; }
        if (!fmt || !std::getline(f, line)) return false;
        int h = 0, w = 0;
        if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) return false;
        std::vector<unsigned char> rgbe((size_t)w * h * 4);
        std::vector<unsigned char> rest((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        size_t pos = 0;
        bool flat = (w < 8 || w >= 32768) || rest.size() < 4 || rest[0] != 2 || rest[1] != 2 || (rest[2] & 0x80);
        if (flat) {
            if (rest.size() < rgbe.size()) return false;
            std::memcpy(rgbe.data(), rest.data(), rgbe.size());
        } else {
            for (int j = 0; j < h; j++) {
                if (pos + 4 > rest.size() || rest[pos] != 2 || rest[pos + 1] != 2 || ((rest[pos + 2] << 8) | rest[pos + 3]) != w) return false;
                pos += 4;
                for (int ch = 0; ch < 4; ch++) {
                    int i = 0;
                    while (i < w) {
                        if (pos >= rest.size()) return false;
                        int n = rest[pos++];
                        if (n > 128) { n -= 128; if (pos >= rest.size() || i + n > w) return false; unsigned char v = rest[pos++]; while (n--) rgbe[((size_t)j * w + i++) * 4 + ch] = v; }
                        else { if (n == 0 || pos + n > rest.size() || i + n > w) return false; while (n--) rgbe[((size_t)j * w + i++) * 4 + ch] = rest[pos++]; }
                    }
                }
            }
        }
        f32.resize((size_t)w * h * 3);
        for (size_t p = 0; p < (size_t)w * h; p++) {
            const unsigned char* q = &rgbe[p * 4];
            if (q[3] != 0) {
                float s = std::ldexp(1.0f, (int)q[3] - (128 + 8));
                f32[p * 3] = q[0] * s; f32[p * 3 + 1] = q[1] * s; f32[p * 3 + 2] = q[2] * s;
            } else f32[p * 3] = f32[p * 3 + 1] = f32[p * 3 + 2] = 0.f;
        }
        width = w; height = h; return true;
    }
};

// ---- material.hpp ---------------------------------------------------------------------------------
class material {
public:
    virtual ~material() = default;
    virtual color emitted(double, double, const point3&) const { return color(0, 0, 0); }
    virtual bool scatter(const ray& r_in, const hit_record& rec, color& attenuation, ray& scattered) const = 0;
    virtual color get_albedo(const hit_record&) const { return color(0, 0, 0); }
    virtual uint32_t flatten(zenith::scene_builder& b) const = 0;
    mutable shared_ptr<zenith::device_object> zr_device_cache_;   // drop-in: the committed scene scatter() runs on
};

class lambertian : public material {
public:
    lambertian(const color& albedo, shared_ptr<texture> bump = nullptr, double strength = 1.0)
        : tex(make_shared<solid_color>(albedo)), bump(bump), strength(strength) {}
    lambertian(shared_ptr<texture> tex, shared_ptr<texture> bump = nullptr, double strength = 1.0) : tex(tex), bump(bump), strength(strength) {}
    bool scatter(const ray& r_in, const hit_record& rec, color& attenuation, ray& scattered) const override { return zenith::device_scatter(*this, r_in, rec, attenuation, scattered); }
    color get_albedo(const hit_record& rec) const override { return tex->value(rec.u, rec.v, rec.p); }
    uint32_t flatten(zenith::scene_builder& b) const override {
        zr_material m{}; m.kind = ZR_MAT_LAMBERTIAN; m.tex = b.texture_id(tex); m.bump_tex = b.texture_id(bump); m.bump_strength = strength;
        return b.add_material(m);
    }
private:
    shared_ptr<texture> tex, bump; double strength;
};

class metal : public material {
public:
    metal(shared_ptr<texture> a, double f, shared_ptr<texture> bump = nullptr, double strength = 1.0)
        : albedo(a), fuzz(f < 1 ? f : 1), bump(bump), strength(strength) {}
    metal(const color& a, double f, shared_ptr<texture> bump = nullptr, double strength = 1.0)
        : albedo(make_shared<solid_color>(a)), fuzz(f < 1 ? f : 1), bump(bump), strength(strength) {}
    bool scatter(const ray& r_in, const hit_record& rec, color& attenuation, ray& scattered) const override { return zenith::device_scatter(*this, r_in, rec, attenuation, scattered); }
    color get_albedo(const hit_record& rec) const override { return albedo->value(rec.u, rec.v, rec.p); }
    uint32_t flatten(zenith::scene_builder& b) const override {
        zr_material m{}; m.kind = ZR_MAT_METAL; m.tex = b.texture_id(albedo); m.bump_tex = b.texture_id(bump); m.bump_strength = strength; m.param = fuzz;
        return b.add_material(m);
    }
private:
    shared_ptr<texture> albedo; double fuzz; shared_ptr<texture> bump; double strength;
};

class dielectric : public material {
public:
    dielectric(double ri, const color& a = color(1.0, 1.0, 1.0)) : ri(ri), albedo(a), bump(nullptr), strength(1.0) {}
    dielectric(double ri, const color& a, shared_ptr<texture> bump, double strength) : ri(ri), albedo(a), bump(bump), strength(strength) {}
    dielectric(double ri, shared_ptr<texture> bump, double strength) : ri(ri), albedo(1.0, 1.0, 1.0), bump(bump), strength(strength) {}
    bool scatter(const ray& r_in, const hit_record& rec, color& attenuation, ray& scattered) const override { return zenith::device_scatter(*this, r_in, rec, attenuation, scattered); }
    color get_albedo(const hit_record&) const override { return color(1.0, 1.0, 1.0); }
    uint32_t flatten(zenith::scene_builder& b) const override {
        zr_material m{}; m.kind = ZR_MAT_DIELECTRIC; m.tex = ZR_NO_TEXTURE; m.bump_tex = b.texture_id(bump); m.bump_strength = strength; m.param = ri;
        m.tint[0] = albedo.x(); m.tint[1] = albedo.y(); m.tint[2] = albedo.z();
        return b.add_material(m);
    }
private:
    double ri; color albedo; shared_ptr<texture> bump; double strength;
};

class diffuse_light : public material {
public:
    diffuse_light(shared_ptr<texture> a) : emit(a) {}
    diffuse_light(color c) : emit(make_shared<solid_color>(c)) {}
    bool scatter(const ray&, const hit_record&, color&, ray&) const override { return false; }
    color emitted(double u, double v, const point3& p) const override { return emit->value(u, v, p); }
    color get_albedo(const hit_record& rec) const override {
        color c = emit->value(rec.u, rec.v, rec.p);
        return color(std::fmin(c.x(), 1.0), std::fmin(c.y(), 1.0), std::fmin(c.z(), 1.0));
    }
    uint32_t flatten(zenith::scene_builder& b) const override {
        zr_material m{}; m.kind = ZR_MAT_LIGHT; m.tex = b.texture_id(emit); m.bump_tex = ZR_NO_TEXTURE;
        return b.add_material(m);
    }
private:
    shared_ptr<texture> emit;
};

// constant_medium.hpp:9-22
class isovolumetric : public material {
public:
    isovolumetric(color c) : tex(make_shared<solid_color>(c)) {}
    isovolumetric(shared_ptr<texture> t) : tex(t) {}
    bool scatter(const ray& r_in, const hit_record& rec, color& attenuation, ray& scattered) const override { return zenith::device_scatter(*this, r_in, rec, attenuation, scattered); }
    uint32_t flatten(zenith::scene_builder& b) const override {
        zr_material m{}; m.kind = ZR_MAT_ISOTROPIC; m.tex = b.texture_id(tex); m.bump_tex = ZR_NO_TEXTURE;
        return b.add_material(m);
    }
private:
    shared_ptr<texture> tex;
};

inline uint32_t zenith::scene_builder::texture_id(const shared_ptr<texture>& t) {
    if (!t) return ZR_NO_TEXTURE;
    auto it = tex_ids.find(t.get());
    if (it != tex_ids.end()) return it->second;
    uint32_t id = t->flatten(*this);
    tex_ids[t.get()] = id;
    return id;
}
inline uint32_t zenith::scene_builder::material_id(const shared_ptr<material>& m) {
    if (!m) return zenith::no_material;
    if (m.get() == last_mat_) return last_mat_id_;   // (a mesh's triangles share one material: no hash lookup per triangle)
    auto it = mat_ids.find(m.get());
    if (it != mat_ids.end()) { last_mat_ = m.get(); last_mat_id_ = it->second; }
    if (it != mat_ids.end()) return it->second;
    uint32_t id = m->flatten(*this);
    mat_ids[m.get()] = id;
    mat_ptrs[id] = m;
    last_mat_ = m.get(); last_mat_id_ = id;
    return id;
}

// ---- hittable_list.hpp ----------------------------------------------------------------------------
class hittable_list : public hittable {
public:
    std::vector<shared_ptr<hittable>> objects;
    hittable_list() {}
    hittable_list(shared_ptr<hittable> o) { add(o); }
    void clear() { objects.clear(); }
    void add(shared_ptr<hittable> o) { objects.push_back(o); bbox = aabb(bbox, o->bounding_box()); }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return bbox; }
    void flatten(zenith::scene_builder& b) const override;   // (defined after `triangle`: long runs of triangles are flattened by all threads)
private:
    aabb bbox;
};

// ---- sphere.hpp / triangle.hpp / cube.hpp -----------------------------------------------------------
class sphere : public hittable {
public:
    sphere(const point3& center, double radius, shared_ptr<material> mat) : center(center), radius_arg(radius), mat(mat) {
        vec3 rv(radius, radius, radius);
        bbox = aabb(center - rv, center + rv);  // from the raw argument, like sphere.hpp:13-14
    }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return bbox; }
    void set_material(shared_ptr<material> m) { mat = m; zr_device_cache_.reset(); }
    void flatten(zenith::scene_builder& b) const override { b.emit_sphere(center, radius_arg, mat); }
private:
    point3 center; double radius_arg; shared_ptr<material> mat; aabb bbox;
};

namespace zenith { template <class Get> size_t flatten_triangle_run(scene_builder& b, size_t n, Get&& get); }
class triangle : public hittable {
public:
    triangle(const point3& a, const point3& b, const point3& c, const vec3& n0, const vec3& n1, const vec3& n2, shared_ptr<material> m)
        : v{a, b, c}, n{n0, n1, n2}, mat(m) {}
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override {  // triangle.hpp:84-101
        double lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            lo[k] = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k])); hi[k] = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
            if (hi[k] - lo[k] < 0.0001) { lo[k] -= 0.0001; hi[k] += 0.0001; }
        }
        return aabb(point3(lo[0], lo[1], lo[2]), point3(hi[0], hi[1], hi[2]));
    }
    void set_material(shared_ptr<material> m) { mat = m; zr_device_cache_.reset(); }
    void flatten(zenith::scene_builder& b) const override { b.emit_triangle(v, n, mat); }
private:
    template <class Get> friend size_t zenith::flatten_triangle_run(zenith::scene_builder&, size_t, Get&&);
    point3 v[3]; vec3 n[3]; shared_ptr<material> mat;
};

// A world is flattened once per render, on the caller's clock (the reference rebuilds its world on every restart, main.cpp:1492-1500):
// a million triangles one virtual call at a time took 0.21-0.25 s.  Runs of at least 16384 consecutive `triangle` objects are
// written by all threads instead (same arrays, same order: entry k of the run is triangle first + k), everything else as before.
// a run of triangle objects through the builder's bulk entry: `get(k)` = object k of n.  Returns how many were taken (0: the
// builder wants them one by one, or the first object is no triangle)
namespace zenith {
template <class Get>
inline size_t flatten_triangle_run(scene_builder& b, size_t n, Get&& get) {
    size_t first_tri = 0, first_obj = 0; uint32_t cf = 0, cc = 0; bool entries = true;
    if (n < 16384) return 0;
    { const hittable& head = get(0); if (typeid(head) != typeid(triangle)) return 0; }
    if (!b.begin_bulk_triangles(n, first_tri, first_obj, cf, cc, entries)) return 0;
    // optimistic: room for all n, every thread copies its share and stops at the first object that is not a triangle; the run ends
    // at the earliest such object and the arrays are cut back to it.  Materials are collected as pointers and turned into ids
    // afterwards, in order (registering a material flattens its textures: not a job for worker threads).
    std::vector<const material*> mp(n);
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t T = std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(16u, hw ? hw : 1u), n / 8192));
    std::vector<size_t> stop(T, n);
    auto work = [&](size_t t, size_t a, size_t e) {
        for (size_t k = a; k < e; k++) {
            const hittable& h = get(k);
            if (typeid(h) != typeid(triangle)) { stop[t] = k; return; }
            const triangle& tr = static_cast<const triangle&>(h);
            double* tv = b.fs.tri_v.data() + (first_tri + k) * 9; double* tn = b.fs.tri_n.data() + (first_tri + k) * 9;
            for (int c = 0; c < 3; c++) { tv[3 * c] = tr.v[c].x(); tv[3 * c + 1] = tr.v[c].y(); tv[3 * c + 2] = tr.v[c].z(); tn[3 * c] = tr.n[c].x(); tn[3 * c + 1] = tr.n[c].y(); tn[3 * c + 2] = tr.n[c].z(); }
            mp[k] = tr.mat.get();
            if (entries) { zr_object o{}; o.type = ZR_PRIM_TRIANGLE; o.index = (uint32_t)(first_tri + k); o.chain_first = cf; o.chain_count = cc; b.fs.objects[first_obj + k] = o; }
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; t++) th.emplace_back(work, t, n * t / T, n * (t + 1) / T);
    work(0, 0, n / T);
    for (auto& x : th) x.join();
    size_t run = n;
    for (size_t t = 0; t < T; t++) if (stop[t] < n) { run = stop[t]; break; }   // (chunks are in order: the first one that stopped ends the run)
    b.end_bulk_triangles(first_tri, run, first_obj + run, entries);
    {   // pointers -> ids, in order; the triangle itself is visited again only when its material has not been seen just before
        const material* last = nullptr; uint32_t last_id = 0; bool have = false;
        for (size_t k = 0; k < run; k++) {
            if (!have || mp[k] != last) { last = mp[k]; last_id = b.material_id(static_cast<const triangle&>(get(k)).mat); have = true; }
            b.fs.tri_mat[first_tri + k] = last_id;
        }
    }
    return run;
}
}  // namespace zenith
inline void hittable_list::flatten(zenith::scene_builder& b) const {
    const size_t N = objects.size();
    b.reserve_hint(N);
    // The bulk path costs O(n) for the n entries it is offered (staging array, array growth, thread start) whatever it takes of them, so it is offered only what
    // it will probably take (offered the whole remaining list after every non-triangle member, a long MIXED list cost O(N^2)): a cheap serial look at
    // the next 16384 entries qualifies a run — a shorter run is flattened one by one without being looked at again — and the window offered grows only while
    // the run keeps filling it.
    size_t i = 0, window = 65536;
    while (i < N) {
        const size_t qmax = std::min<size_t>(N - i, 16384);
        size_t q = 0;
        for (; q < qmax; q++) { const hittable& h = *objects[i + q]; if (typeid(h) != typeid(triangle)) break; }
        size_t took = 0;
        if (q == 16384) {
            const size_t n = std::min(N - i, window);
            took = zenith::flatten_triangle_run(b, n, [&](size_t k) -> const hittable& { return *objects[i + k]; });
            window = (took == n) ? std::min<size_t>(window * 4, N) : 65536;
        }
        if (took) { i += took; continue; }
        const size_t one_by_one = std::max<size_t>(q, 1);   // a short run of triangles and / or one other member (or a builder that wants them one by one)
        for (size_t k = 0; k < one_by_one; k++) objects[i + k]->flatten(b);
        i += one_by_one;
    }
}

class cube : public hittable {
public:
    cube(const point3& mn, const point3& mx, shared_ptr<material> mat) : mat(mat), min_p(mn), max_p(mx) {
        half = 0.5 * (mx - mn); center = mn + half;
    }
    cube(const point3& c, shared_ptr<material> mat) : half(1.0, 1.0, 1.0), center(c), mat(mat) { min_p = c - half; max_p = c + half; }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override {
        return aabb(interval(min_p.x(), max_p.x()).expand(0.0001), interval(min_p.y(), max_p.y()).expand(0.0001),
                    interval(min_p.z(), max_p.z()).expand(0.0001));
    }
    void set_material(shared_ptr<material> m) { mat = m; zr_device_cache_.reset(); }
    void flatten(zenith::scene_builder& b) const override { b.emit_cube(half, center, min_p, max_p, mat); }
private:
    vec3 half; point3 center; shared_ptr<material> mat; point3 min_p, max_p;
};

// ---- constant_medium.hpp ----------------------------------------------------------------------------
class constant_medium : public hittable {
public:
    constant_medium(shared_ptr<hittable> boundary, double density, shared_ptr<texture> tex)
        : boundary(boundary), density(density), phase(make_shared<isovolumetric>(tex)) {}
    constant_medium(shared_ptr<hittable> boundary, double density, color c)
        : boundary(boundary), density(density), phase(make_shared<isovolumetric>(c)) {}
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return boundary->bounding_box(); }
    void flatten(zenith::scene_builder& b) const override { b.emit_medium(*boundary, density, b.material_id(phase)); }
private:
    shared_ptr<hittable> boundary; double density; shared_ptr<material> phase;
};

inline void zenith::scene_builder::emit_medium(const hittable& boundary, double density, uint32_t iso_material) {
    if (in_boundary) { unsupported("constant_medium used as the boundary of another medium"); return; }
    in_boundary = true; boundary_base = chain.size(); captured_ok = false;
    size_t objects_before = fs.objects.size();
    boundary.flatten(*this);
    in_boundary = false;
    if (!captured_ok || fs.objects.size() != objects_before || (captured.type != ZR_PRIM_SPHERE && captured.type != ZR_PRIM_CUBE)) {
        unsupported("constant_medium boundary must be one (optionally wrapped) sphere or cube");
        return;
    }
    zr_medium m{};
    m.boundary_type = captured.type; m.boundary_index = captured.index;
    m.chain_first = captured.chain_first; m.chain_count = captured.chain_count;
    m.mat = iso_material; m.neg_inv_density = -1.0 / density;
    fs.media.push_back(m);
    emit(ZR_PRIM_MEDIUM, (uint32_t)fs.media.size() - 1);
}

// ---- wrappers: translate / rotate_* / scale / material_instance --------------------------------------
namespace zenith {
inline aabb rotated_box(const aabb& in, int axis, double s, double c) {
    point3 lo(infinity, infinity, infinity), hi(-infinity, -infinity, -infinity);
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
        double x = i * in.x.max + (1 - i) * in.x.min, y = j * in.y.max + (1 - j) * in.y.min, z = k * in.z.max + (1 - k) * in.z.min;
        vec3 t(x, y, z);
        if (axis == 1) { t[0] = c * x - s * z; t[2] = s * x + c * z; }  // the true map; rotate_y.hpp:26-27 uses the inverse
        else if (axis == 0) { t[1] = c * y - s * z; t[2] = s * y + c * z; }
        else { t[0] = c * x - s * y; t[1] = s * x + c * y; }
        for (int q = 0; q < 3; q++) { lo[q] = std::fmin(lo[q], t[q]); hi[q] = std::fmax(hi[q], t[q]); }
    }
    return aabb(lo, hi);
}
}  // namespace zenith

class translate : public hittable {
public:
    translate(shared_ptr<hittable> p, const vec3& displacement) : ptr(p), offset(displacement) { bbox = ptr->bounding_box() + offset; }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return bbox; }
    void flatten(zenith::scene_builder& b) const override { b.push_op(ZR_OP_TRANSLATE, offset.x(), offset.y(), offset.z()); ptr->flatten(b); b.pop_op(); }
private:
    shared_ptr<hittable> ptr; vec3 offset; aabb bbox;
};

class rotate_y : public hittable {  // angle in RADIANS (rotate_y.hpp:9-12)
public:
    rotate_y(shared_ptr<hittable> p, double angle_rad) : ptr(p), s(std::sin(angle_rad)), c(std::cos(angle_rad)) {
        bbox = zenith::rotated_box(ptr->bounding_box(), 1, s, c);
    }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return bbox; }
    void flatten(zenith::scene_builder& b) const override { b.push_op(ZR_OP_ROTATE_Y, s, c, 0); ptr->flatten(b); b.pop_op(); }
private:
    shared_ptr<hittable> ptr; double s, c; aabb bbox;
};

class rotate_x : public hittable {  // angle in DEGREES (rotate_x.hpp:9-12)
public:
    rotate_x(shared_ptr<hittable> p, double angle) : ptr(p) {
        double r = degrees_to_radians(angle); s = std::sin(r); c = std::cos(r);
        bbox = zenith::rotated_box(ptr->bounding_box(), 0, s, c);
    }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return bbox; }
    void flatten(zenith::scene_builder& b) const override { b.push_op(ZR_OP_ROTATE_X, s, c, 0); ptr->flatten(b); b.pop_op(); }
private:
    shared_ptr<hittable> ptr; double s, c; aabb bbox;
};

class rotate_z : public hittable {  // angle in DEGREES (rotate_z.hpp:9-12)
public:
    rotate_z(shared_ptr<hittable> p, double angle) : ptr(p) {
        double r = degrees_to_radians(angle); s = std::sin(r); c = std::cos(r);
        bbox = zenith::rotated_box(ptr->bounding_box(), 2, s, c);
    }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return bbox; }
    void flatten(zenith::scene_builder& b) const override { b.push_op(ZR_OP_ROTATE_Z, s, c, 0); ptr->flatten(b); b.pop_op(); }
private:
    shared_ptr<hittable> ptr; double s, c; aabb bbox;
};

class scale : public hittable {
public:
    scale(shared_ptr<hittable> object, const vec3& factors) : object(object), s(factors) {
        aabb in = object->bounding_box();
        bbox = aabb(point3(in.x.min * s.x(), in.y.min * s.y(), in.z.min * s.z()), point3(in.x.max * s.x(), in.y.max * s.y(), in.z.max * s.z()));
    }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return bbox; }
    void flatten(zenith::scene_builder& b) const override { b.push_op(ZR_OP_SCALE, s.x(), s.y(), s.z()); object->flatten(b); b.pop_op(); }
private:
    shared_ptr<hittable> object; vec3 s; aabb bbox;
};

class material_instance : public hittable {
public:
    material_instance(shared_ptr<hittable> obj, shared_ptr<material> mat) : object(obj), new_material(mat) {}
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return object->bounding_box(); }
    void set_material(shared_ptr<material> m) { new_material = m; zr_device_cache_.reset(); }
    void flatten(zenith::scene_builder& b) const override {
        // a null material falls back to magenta lambertian (material_instance.hpp:22-26)
        static const shared_ptr<material> error_mat = make_shared<lambertian>(color(1, 0, 1));
        b.push_op(ZR_OP_MATERIAL, 0, 0, 0, b.material_id(new_material ? new_material : error_mat));
        object->flatten(b);
        b.pop_op();
    }
private:
    shared_ptr<hittable> object; shared_ptr<material> new_material;
};

// ---- bvh.hpp: the tree itself is built on the device side of the ABI (zr_scene_commit) ---------------
class bvh_node : public hittable {
public:
    bvh_node(hittable_list l) : list(std::move(l)) { zenith::consume_draws(zenith::bvh_ctor_draws(list.objects.size())); }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return list.bounding_box(); }
    void flatten(zenith::scene_builder& b) const override { b.emit_run(this, [&] { list.flatten(b); }); }
private:
    hittable_list list;
};


// ---- model.hpp: OBJ mesh -> recentred, scaled triangles ------------------------------------------------
// model(filename, mat, scale): /root/reference/model.hpp:12-103.  The reference parses with the vendored
// tiny_obj_loader (float attributes, polygons triangulated); the drop-in has its own minimal Wavefront reader:
// `v x y z [r g b]`, `vn`, `f` with v, v/vt, v//vn, v/vt/vn and negative (relative) indices.  Triangulation follows
// tinyobj 2.0: quads are split along the SHORTER diagonal, compared in float ([0,1,2][0,2,3] if |v2-v0|^2 < |v3-v1|^2,
// else [0,1,3][1,2,3]); polygons with more than four vertices are fanned (tinyobj ear-clips them: documented
// deviation, the reference's assets contain triangles and quads only).
// Semantics kept: the bounding box is taken over ALL `v` lines; x/z are centred on it and y-min moves to 0
// (model.hpp:23-42); vertex = (v - offset) * scale in double (model.hpp:47-53); a face uses its vn normals when
// its first corner has one, otherwise one flat normal unit(cross(v1-v0, v2-v0)) (model.hpp:70-85).
class model : public hittable {
public:
    model(const std::string& filename, shared_ptr<material> mat, double scale = 1.0) : mat(mat) {
        std::vector<float> V, N;
        struct corner { int v, n; };
        std::vector<std::vector<corner>> faces;
        std::ifstream f(filename);
        if (!f) { std::cerr << "Cannot load the model: " << filename << std::endl; return; }
        std::string line;
        while (std::getline(f, line)) {
            const char* p = line.c_str();
            while (*p == ' ' || *p == '\t') p++;
            if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
                char* e; p += 2;
                for (int k = 0; k < 3; k++) { V.push_back((float)std::strtod(p, &e)); p = e; }
            } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
                char* e; p += 3;
                for (int k = 0; k < 3; k++) { N.push_back((float)std::strtod(p, &e)); p = e; }
            } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
                p += 2;
                std::vector<corner> fc;
                for (;;) {
                    while (*p == ' ' || *p == '\t') p++;
                    if (*p == 0 || *p == '\r' || *p == '\n' || *p == '#') break;
                    char* e;
                    long vi = std::strtol(p, &e, 10), ni = 0; bool has_n = false;
                    if (e == p) break;
                    p = e;
                    if (*p == '/') {
                        p++;
                        if (*p != '/') { (void)std::strtol(p, &e, 10); p = e; }   // vt
                        if (*p == '/') { p++; ni = std::strtol(p, &e, 10); has_n = e != p; p = e; }
                    }
                    corner c;
                    c.v = vi > 0 ? (int)vi - 1 : (int)(V.size() / 3) + (int)vi;
                    c.n = !has_n ? -1 : (ni > 0 ? (int)ni - 1 : (int)(N.size() / 3) + (int)ni);
                    fc.push_back(c);
                }
                if (fc.size() >= 3) faces.push_back(fc);
            }
        }
        double lo[3] = {infinity, infinity, infinity}, hi[3] = {-infinity, -infinity, -infinity};
        for (size_t i = 0; i + 2 < V.size(); i += 3)
            for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], V[i + k]); hi[k] = std::fmax(hi[k], V[i + k]); }
        const vec3 off((lo[0] + hi[0]) / 2.0, lo[1], (lo[2] + hi[2]) / 2.0);
        auto vert = [&](int i) { return point3((V[3 * i + 0] - off.x()) * scale, (V[3 * i + 1] - off.y()) * scale, (V[3 * i + 2] - off.z()) * scale); };
        auto norm = [&](int i) { return vec3(N[3 * i + 0], N[3 * i + 1], N[3 * i + 2]); };
        auto valid = [&](const corner& c) { return c.v >= 0 && (size_t)(3 * c.v + 2) < V.size(); };
        auto emit = [&](const corner& a, const corner& b, const corner& c) {
            if (!valid(a) || !valid(b) || !valid(c)) return;
            point3 v0 = vert(a.v), v1 = vert(b.v), v2 = vert(c.v);
            vec3 n0, n1, n2;
            if (a.n >= 0 && b.n >= 0 && c.n >= 0 && (size_t)(3 * std::max(a.n, std::max(b.n, c.n)) + 2) < N.size()) { n0 = norm(a.n); n1 = norm(b.n); n2 = norm(c.n); }
            else n0 = n1 = n2 = unit_vector(cross(v1 - v0, v2 - v0));
            tris.push_back(make_shared<triangle>(v0, v1, v2, n0, n1, n2, mat));
            bbox = aabb(bbox, tris.back()->bounding_box());
        };
        for (const auto& fc : faces) {
            if (fc.size() == 3) emit(fc[0], fc[1], fc[2]);
            else if (fc.size() == 4 && valid(fc[0]) && valid(fc[1]) && valid(fc[2]) && valid(fc[3])) {
                const float* a = &V[3 * fc[0].v]; const float* b = &V[3 * fc[1].v]; const float* c = &V[3 * fc[2].v]; const float* d = &V[3 * fc[3].v];
                const float e02x = c[0] - a[0], e02y = c[1] - a[1], e02z = c[2] - a[2];
                const float e13x = d[0] - b[0], e13y = d[1] - b[1], e13z = d[2] - b[2];
                const float s02 = e02x * e02x + e02y * e02y + e02z * e02z, s13 = e13x * e13x + e13y * e13y + e13z * e13z;
                if (s02 < s13) { emit(fc[0], fc[1], fc[2]); emit(fc[0], fc[2], fc[3]); }
                else { emit(fc[0], fc[1], fc[3]); emit(fc[1], fc[2], fc[3]); }
            } else {
                for (size_t k = 1; k + 1 < fc.size(); k++) emit(fc[0], fc[k], fc[k + 1]);
            }
        }
        zenith::consume_draws(zenith::bvh_ctor_draws(tris.size()));   // mesh_bvh = make_shared<bvh_node>(triangles), model.hpp:95
        std::cout << "Model: " << filename << " loaded (" << tris.size() << " triangles)." << std::endl;
    }
    bool hit(const ray& r, interval ray_t, hit_record& rec, int = 0, bool = false) const override { return zenith::device_hit(*this, r, ray_t, rec); }
    aabb bounding_box() const override { return bbox; }
    void set_material(std::shared_ptr<material> m) { mat = m; }   // like the reference's: does not re-material existing triangles
    void flatten(zenith::scene_builder& b) const override {
        b.emit_run(this, [&] {
            size_t i = 0;
            while (i < tris.size()) {
                const size_t took = zenith::flatten_triangle_run(b, tris.size() - i, [&](size_t k) -> const hittable& { return *tris[i + k]; });
                if (took) { i += took; continue; }
                tris[i]->flatten(b); i++;
            }
        });
    }
    size_t triangle_count() const { return tris.size(); }
private:
    std::vector<shared_ptr<triangle>> tris;
    shared_ptr<material> mat;
    aabb bbox;
};

// ---- environment.hpp ---------------------------------------------------------------------------------
inline const std::string HDR_DIR = "assets/hdr_maps/";
struct EnvironmentSettings {
    enum Mode { PHYSICAL_SUN, HDR_MAP, SOLID_COLOR };
    Mode _mode = PHYSICAL_SUN;
    bool needs_ui_sync = false;
    Mode mode() const { return _mode; }
    void set_mode(Mode m) { if (_mode != m) { _mode = m; needs_ui_sync = true; } }
    color background_color = color(0.0, 0.0, 0.0);
    std::string current_hdr_name = "None";
    std::string current_hdr_path = HDR_DIR;
    double intensity = 1.0;
    double hdri_rotation = 0.0, hdri_tilt = 0.0, hdri_roll = 0.0;
    shared_ptr<image_texture> hdr_texture = nullptr;
    void load_hdr(const std::string& path) {
        if (path.empty()) { set_mode(SOLID_COLOR); background_color = color(0, 0, 0); current_hdr_name = "None (Black)"; return; }
        hdr_texture = make_shared<image_texture>(path.c_str(), true);
        size_t slash = path.find_last_of("/\\");
        current_hdr_name = slash == std::string::npos ? path : path.substr(slash + 1);
        current_hdr_path = path;
        set_mode(HDR_MAP);
    }
    vec3 sun_direction = unit_vector(vec3(1.0, 0.5, -0.5));
    color sun_color = color(1.0, 1.0, 1.0);
    bool auto_sun_color = true;
    double sun_intensity = 1.0;
    double sun_size = 1.0;
};

// post_processor (color_processing.hpp:44-75): the reference's public fields and defaults.  The arithmetic (process, bloom,
// sharpening, analyze_framebuffer) runs on the device: camera::process_framebuffer / camera::analyze below.
struct debug_flags {
    bool red = false, green = false, blue = false, luminance = false, bvh = false;
    bool any_active() const { return red || green || blue || luminance || bvh; }
};
struct image_statistics {
    float average_luminance = 0.0f, max_luminance = 0.0f;
    int histogram[256] = {0};
    float normalized_histogram[256] = {0.0f};
    void normalize() {  // color_processing.hpp:15-27
        int max_pixels = 0;
        for (int i = 0; i < 256; i++) if (histogram[i] > max_pixels) max_pixels = histogram[i];
        if (max_pixels > 0) for (int i = 0; i < 256; i++) normalized_histogram[i] = static_cast<float>(histogram[i]) / max_pixels;
    }
};
struct post_processor {
    mutable float exposure = 0.5f;
    float saturation = 1.0f, contrast = 1.0f, hue_shift = 0.0f, vignette_intensity = 1.0f;
    vec3 color_balance = vec3(1.0f, 1.0f, 1.0f);
    float z_depth_max_dist = 1.0f;
    float exposure_compensation_stops = 0.0f;
    bool use_aces_tone_mapping = false, use_auto_exposure = false;
    float target_luminance = 0.12f;
    mutable debug_flags debug;
    bool use_bloom = false;
    float bloom_threshold = 1.0f, bloom_intensity = 0.3f;
    int bloom_radius = 4;
    bool needs_update = true;
    mutable image_statistics last_stats;
    bool use_sharpening = false;
    double sharpen_amount = 0.2;

    double apply_auto_exposure(const image_statistics& stats) const {  // color_processing.hpp:186-205 (scalar, host)
        if (stats.average_luminance <= 0.0) return static_cast<double>(exposure);
        if (!use_auto_exposure) return std::clamp(static_cast<float>(exposure), 0.01f, 10.0f);
        double safe_luminance = std::max(static_cast<double>(stats.average_luminance), 0.02);
        double raw_exposure = target_luminance / safe_luminance;
        double current_exp = raw_exposure * std::pow(2.0, static_cast<double>(exposure_compensation_stops));
        return std::clamp(static_cast<float>(current_exp), 0.01f, 4.0f);
    }
    zr_post_params to_zr() const {
        zr_post_params p{};
        p.exposure = exposure; p.saturation = saturation; p.contrast = contrast; p.hue_shift = hue_shift; p.vignette_intensity = vignette_intensity;
        p.bloom_threshold = bloom_threshold; p.bloom_intensity = bloom_intensity; p.bloom_radius = bloom_radius;
        for (int k = 0; k < 3; k++) p.color_balance[k] = color_balance[k];
        p.sharpen_amount = sharpen_amount; p.use_aces_tone_mapping = use_aces_tone_mapping; p.use_bloom = use_bloom; p.use_sharpening = use_sharpening;
        p.debug_red = debug.red; p.debug_green = debug.green; p.debug_blue = debug.blue; p.debug_luminance = debug.luminance; p.debug_bvh = debug.bvh;
        return p;
    }
};

namespace zenith {
inline zr_env to_zr_env(const EnvironmentSettings& e, scene_builder& b) {
    zr_env z{};
    z.mode = (uint32_t)e._mode;
    z.hdr_texture = e.hdr_texture ? b.texture_id(std::static_pointer_cast<texture>(e.hdr_texture)) : ZR_NO_TEXTURE;
    for (int k = 0; k < 3; k++) { z.background_color[k] = e.background_color[k]; z.sun_direction[k] = e.sun_direction[k]; z.sun_color[k] = e.sun_color[k]; }
    z.intensity = e.intensity; z.hdri_rotation = e.hdri_rotation; z.hdri_tilt = e.hdri_tilt; z.hdri_roll = e.hdri_roll;
    z.sun_intensity = e.sun_intensity; z.sun_size = e.sun_size;
    return z;
}
// Device contexts are kept in ONE process-wide pool, per device ordinal, and LEASED for the duration of a call: the reference
// starts a fresh thread for every render (main.cpp:1520-1531), and a context per thread would create — and free — the
// pipeline's buffers (tens of GB for a 1080p frame at 512 spp) on every restart.  A lease is exclusive (a zr_ctx serves one
// caller at a time); threads that render one after the other reuse the same context, threads that render at the same time get
// one each.  The pool lives as long as the process.
struct context_pool {
    struct entry { zr_ctx* c; int device; bool busy; };
    std::mutex m;
    std::condition_variable cv;
    std::vector<entry> all;
    ~context_pool() { for (entry& e : all) if (e.c) zr_destroy(e.c); }
    zr_ctx* acquire(int device) {
        {
            std::lock_guard<std::mutex> lk(m);
            for (entry& e : all) if (e.device == device && !e.busy) { e.busy = true; return e.c; }
        }
        zr_ctx* c = zr_create(device);   // (outside the lock: creating a context takes milliseconds)
        if (!c) return nullptr;
        std::lock_guard<std::mutex> lk(m);
        all.push_back(entry{c, device, true});
        return c;
    }
    void acquire(zr_ctx* c) {   // a particular context (the one a cached scene was committed on): wait until it is free
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { for (entry& e : all) if (e.c == c) return !e.busy; return true; });
        for (entry& e : all) if (e.c == c) e.busy = true;
    }
    void release(zr_ctx* c) {
        { std::lock_guard<std::mutex> lk(m); for (entry& e : all) if (e.c == c) e.busy = false; }
        cv.notify_all();
    }
    size_t created() { std::lock_guard<std::mutex> lk(m); return all.size(); }
};
inline context_pool& contexts() { static context_pool p; return p; }
// contexts created by this process so far (tests: three renders from three successive threads create one)
inline size_t contexts_created() { return contexts().created(); }
struct context_lease {
    zr_ctx* ctx;
    explicit context_lease(int device) : ctx(contexts().acquire(device)) {}
    explicit context_lease(zr_ctx* c) : ctx(c) { if (c) contexts().acquire(c); }
    context_lease(const context_lease&) = delete;
    context_lease& operator=(const context_lease&) = delete;
    ~context_lease() { if (ctx) contexts().release(ctx); }
    operator zr_ctx*() const { return ctx; }
};
// Flattened worlds are leased from a process-wide stash as well: a camera flattens its world on every render, and reusing the
// arrays of an earlier render saves touching a quarter of a gigabyte of fresh pages for a million triangles.
struct flat_stash { std::mutex m; std::vector<std::unique_ptr<flat_scene>> free; };
inline flat_stash& flat_scenes() { static flat_stash s; return s; }
struct flat_lease {
    std::unique_ptr<flat_scene> fs;
    flat_lease() {
        { std::lock_guard<std::mutex> lk(flat_scenes().m); if (!flat_scenes().free.empty()) { fs = std::move(flat_scenes().free.back()); flat_scenes().free.pop_back(); } }
        if (!fs) fs.reset(new flat_scene()); else fs->clear();
    }
    flat_lease(const flat_lease&) = delete;
    flat_lease& operator=(const flat_lease&) = delete;
    ~flat_lease() { std::lock_guard<std::mutex> lk(flat_scenes().m); if (flat_scenes().free.size() < 2) flat_scenes().free.push_back(std::move(fs)); }
};
// the device hit() / scatter() calls of scene objects use: the one of the camera that rendered last (camera::device), 0 before that
inline std::atomic<int>& object_device() { static std::atomic<int> d{0}; return d; }

// ---- callable hit() / scatter(): one ray through the device ---------------------------------------------------------
// The object is flattened once and committed as a scene of its own (cached in the object, per context; set_material drops
// the cache, any other mutation of a child after the first call needs zenith::forget(obj)).  hit() = zr_trace over
// interval [ray_t.min, ray_t.max] with each primitive's own reading of the bounds; scatter() = zr_kat_scatter with the
// draws taken from the thread's random_double() stream, which is advanced by the number of draws the material made —
// i.e. the call consumes random_double() exactly as the reference's scatter would (common.hpp:29-34, material.hpp).
// A constant_medium draws its distance from the off-stream medium key of the current stream position (zr_rng.h).
struct device_object {
    zr_ctx* ctx = nullptr; zr_scene* sc = nullptr; int device = 0;
    flat_scene fs;
    std::unordered_map<uint32_t, shared_ptr<material>> mats;
    ~device_object() { if (sc) zr_scene_destroy(sc); }   // (frees the scene's device arrays; touches no context state: no lease)
};
template <class Flatten>
inline shared_ptr<device_object> device_commit(shared_ptr<device_object>& cache, const char* what, Flatten&& flatten) {
    if (cache && cache->device == object_device().load()) return cache;   // committed before, on a context of this device (which the pool keeps alive)
    context_lease lease(object_device().load());
    zr_ctx* ctx = lease;
    if (!ctx) throw std::runtime_error(std::string(what) + ": no device context: " + zr_last_error());
    auto d = make_shared<device_object>();
    d->ctx = ctx; d->device = object_device().load();
    scene_builder b(d->fs);
    flatten(b);
    b.finish();
    d->mats = b.mat_ptrs;
    for (const auto& w : d->fs.warnings) std::cerr << "[zenith] " << w << "\n";
    d->sc = zr_scene_create(ctx);
    zr_scene_desc desc = d->fs.desc();
    if (!d->sc || zr_scene_set_all_borrowed(d->sc, &desc) != ZR_OK || zr_scene_commit(d->sc) != ZR_OK)
        throw std::runtime_error(std::string(what) + ": " + zr_last_error());
    cache = d;
    return d;
}
inline void forget(const hittable& h) { h.zr_device_cache_.reset(); }
inline void forget(const material& m) { m.zr_device_cache_.reset(); }

inline bool device_hit(const hittable& self, const ray& r, const interval& ray_t, hit_record& rec) {
    auto d = device_commit(self.zr_device_cache_, "hittable::hit", [&](scene_builder& b) { self.flatten(b); });
    const double rays6[6] = {r.origin().x(), r.origin().y(), r.origin().z(), r.direction().x(), r.direction().y(), r.direction().z()};
    host_rng& g = rng_state();
    zr_hit h{};
    context_lease lease(d->ctx);   // the object's scene lives on that context: exclusive for the call
    // stream key of this call = (host key, host draw position, 0): a medium's off-stream draw changes as the host stream advances
    if (zr_trace(d->ctx, d->sc, rays6, 1, ray_t.min, ray_t.max, g.key, g.k, 0, &h) != ZR_OK) throw std::runtime_error(std::string("hittable::hit: ") + zr_last_error());
    if (!d->fs.media.empty()) g.k++;   // constant_medium::hit consumes a draw (constant_medium.hpp:64)
    if (h.mat == 0xFFFFFFFFu && h.t == 0.0) return false;
    rec.p = point3(h.p[0], h.p[1], h.p[2]); rec.normal = vec3(h.normal[0], h.normal[1], h.normal[2]);
    rec.tangent = vec3(h.tangent[0], h.tangent[1], h.tangent[2]); rec.bitangent = vec3(h.bitangent[0], h.bitangent[1], h.bitangent[2]);
    rec.t = h.t; rec.u = h.u; rec.v = h.v; rec.front_face = h.front_face != 0;
    auto it = d->mats.find(h.mat);
    rec.mat = it != d->mats.end() ? it->second : nullptr;
    return true;
}

inline bool device_scatter(const material& self, const ray& r_in, const hit_record& rec, vec3& attenuation, ray& scattered) {
    uint32_t id = 0;
    auto d = device_commit(self.zr_device_cache_, "material::scatter", [&](scene_builder& b) { id = self.flatten(b); });
    id = (uint32_t)d->fs.materials.size() - 1;   // a material flattens its textures first and itself last
    const double rays6[6] = {r_in.origin().x(), r_in.origin().y(), r_in.origin().z(), r_in.direction().x(), r_in.direction().y(), r_in.direction().z()};
    zr_hit h{};
    for (int k = 0; k < 3; k++) { h.p[k] = rec.p[k]; h.normal[k] = rec.normal[k]; h.tangent[k] = rec.tangent[k]; h.bitangent[k] = rec.bitangent[k]; }
    h.t = rec.t; h.u = rec.u; h.v = rec.v; h.mat = id; h.front_face = rec.front_face ? 1u : 0u;
    host_rng& g = rng_state();
    zr_scatter_out o{};
    context_lease lease(d->ctx);
    if (zr_kat_scatter(d->ctx, d->sc, rays6, &h, &g.key, &g.k, 1, &o) != ZR_OK) throw std::runtime_error(std::string("material::scatter: ") + zr_last_error());
    g.k += o.draws;
    if (!o.scattered) return false;
    attenuation = vec3(o.attenuation[0], o.attenuation[1], o.attenuation[2]);
    scattered = ray(point3(o.origin[0], o.origin[1], o.origin[2]), vec3(o.direction[0], o.direction[1], o.direction[2]));
    return true;
}
}  // namespace zenith

// ---- camera.hpp: public configuration + render() ------------------------------------------------------
class camera {
public:
    double aspect_ratio = 1.0;
    int image_width = 400, image_height = 225;
    int samples_per_pixel = 30;
    int current_samples_count = 0;
    int max_depth = 10;
    double sky_intesity = 1.0;
    double vfov = 30;
    point3 lookfrom = point3(10, 1.5, 0), lookat = point3(0, 0, 0);
    vec3 vup = vec3(0, 1, 0);
    double defocus_angle = 0.5, focus_dist = 10;
    bool use_denoiser = false;
    bool use_albedo_buffer = false, use_normal_buffer = false, use_z_depth_buffer = false, use_reflection = false, use_refraction = false;
    std::vector<color> render_accumulator;
    std::vector<color> albedo_buffer, normal_buffer, z_depth_buffer;  // first-hit passes (camera.hpp:81-83), filled when their flag is set
    std::vector<color> reflection_buffer, refraction_buffer;          // second-path split (camera.hpp:84-85, 490-517), filled when either flag is set
    std::atomic<int> lines_rendered{0};
    uint64_t seed = 0x5EED0000ull;  // extension: the reference cannot be seeded (common.hpp:30-31)
    int device = 0;                 // extension: HIP device ordinal
    zr_counters last_counters{};

    void reset_accumulator() {  // camera.hpp:209-233
        render_accumulator.assign((size_t)image_width * image_height, color(0, 0, 0));
        albedo_buffer.assign(render_accumulator.size(), color(0, 0, 0));
        normal_buffer.assign(render_accumulator.size(), color(0, 0, 0));
        z_depth_buffer.assign(render_accumulator.size(), color(0, 0, 0));
        reflection_buffer.assign(render_accumulator.size(), color(0, 0, 0));
        refraction_buffer.assign(render_accumulator.size(), color(0, 0, 0));
        current_samples_count = 0; lines_rendered = 0;
    }

    // camera::process_framebuffer_to_image up to the PNG encoder (camera.hpp:701-780): bloom, sharpening, exposure,
    // post_processor::process, 8-bit RGB — on the device, byte-exact with the reference.  Data passes (albedo, normals, z-depth,
    // reflection, refraction) are clamped and gamma-corrected only, as save_render_pass asks (camera.hpp:314-345).
    bool process_framebuffer(const std::vector<color>& buffer, const post_processor& pp, std::vector<unsigned char>& rgb8,
                             bool is_data_pass = false, bool apply_gamma = true) const {
        rgb8.assign((size_t)image_width * image_height * 3, 0);
        zenith::context_lease lease(device);
        zr_ctx* ctx = lease;
        if (!ctx || buffer.size() != (size_t)image_width * image_height) { std::cerr << "[zenith] process_framebuffer: " << (ctx ? "buffer size mismatch" : zr_last_error()) << "\n"; return false; }
        zr_post_params p = pp.to_zr();
        int rc = zr_post_process(ctx, &p, reinterpret_cast<const double*>(buffer.data()), image_width, image_height, is_data_pass, apply_gamma, rgb8.data());
        if (rc != ZR_OK) std::cerr << "[zenith] process_framebuffer failed: " << zr_last_error() << "\n";
        return rc == ZR_OK;
    }
    // post_processor::analyze_framebuffer (color_processing.hpp:150-183) on the device
    image_statistics analyze(const std::vector<color>& buffer) const {
        image_statistics st;
        zenith::context_lease lease(device);
        zr_ctx* ctx = lease;
        zr_image_stats z{};
        if (ctx && !buffer.empty() && zr_analyze_frame(ctx, reinterpret_cast<const double*>(buffer.data()), buffer.size(), &z) == ZR_OK) {
            st.average_luminance = z.average_luminance; st.max_luminance = z.max_luminance;
            for (int i = 0; i < 256; i++) st.histogram[i] = z.histogram[i];
            st.normalize();
        }
        return st;
    }

    // camera.hpp:236.  Blocking; fills render_accumulator (mean radiance, row-major, idx = j*W + i)."
    void render(const hittable& world, const EnvironmentSettings& env, const post_processor& post, std::atomic<bool>& render_flag) {
        static_assert(sizeof(std::atomic<bool>) == 1 && sizeof(std::atomic<int>) == sizeof(int), "flag layout");
        if (image_width < 1) image_width = 1;
        if (image_height < 1) image_height = 1;
        aspect_ratio = double(image_width) / image_height;
        lines_rendered = 0;
        render_accumulator.assign((size_t)image_width * image_height, color(0, 0, 0));  // the reference only fills (camera.hpp:420)
        const bool stats = std::getenv("ZR_COMMIT_STATS") != nullptr;
        auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_ph = now_s();
        auto ph = [&](const char* what) { if (stats) { const double t = now_s(); std::fprintf(stderr, "[zenith] render: %-20s %.1f ms\n", what, (t - t_ph) * 1e3); t_ph = t; } };
        zenith::flat_lease lease_fs;                 // (the arrays of an earlier render of this process are reused: no fresh pages to touch)
        zenith::flat_scene& fs = *lease_fs.fs;
        zenith::scene_builder b(fs);
        world.flatten(b);
        b.finish();
        ph("flatten");
        zr_env zenv = zenith::to_zr_env(env, b);
        for (const auto& w : fs.warnings) std::cerr << "[zenith] " << w << "\n";
        zenith::object_device() = device;
        int rc;
        {
        zenith::context_lease lease(device);   // one context of the process-wide pool, exclusively, for this frame
        zr_ctx* ctx = lease;
        if (!ctx) { std::cerr << "[zenith] render failed: " << zr_last_error() << "\n"; return; }
        zr_scene* sc = zr_scene_create(ctx);
        zr_scene_desc d = fs.desc();
        rc = sc ? zr_scene_set_all_borrowed(sc, &d) : ZR_E_DEVICE;
        if (rc == ZR_OK) rc = zr_scene_commit(sc);
        ph("commit");
        if (rc == ZR_OK) {
            zr_camera zc{};
            zc.image_width = image_width; zc.image_height = image_height; zc.samples_per_pixel = samples_per_pixel; zc.max_depth = max_depth;
            zc.vfov = vfov; zc.defocus_angle = defocus_angle; zc.focus_dist = focus_dist;
            for (int k = 0; k < 3; k++) { zc.lookfrom[k] = lookfrom[k]; zc.lookat[k] = lookat[k]; zc.vup[k] = vup[k]; }
            rc = zr_render(ctx, sc, &zc, &zenv, seed, nullptr, 0, reinterpret_cast<double*>(render_accumulator.data()),
                           reinterpret_cast<volatile const uint8_t*>(&render_flag), reinterpret_cast<volatile int*>(&lines_rendered));
            zr_get_counters(ctx, &last_counters);
            ph("zr_render");
            if (rc == ZR_OK && (use_albedo_buffer || use_normal_buffer || use_z_depth_buffer)) {
                const size_t npx = (size_t)image_width * image_height;
                if (use_albedo_buffer) albedo_buffer.assign(npx, color(0, 0, 0));
                if (use_normal_buffer) normal_buffer.assign(npx, color(0, 0, 0));
                if (use_z_depth_buffer) z_depth_buffer.assign(npx, color(0, 0, 0));
                zr_aov_params ap{post.z_depth_max_dist};
                rc = zr_render_aov(ctx, sc, &zc, seed, nullptr, &ap, use_albedo_buffer ? reinterpret_cast<double*>(albedo_buffer.data()) : nullptr,
                                   use_normal_buffer ? reinterpret_cast<double*>(normal_buffer.data()) : nullptr,
                                   use_z_depth_buffer ? reinterpret_cast<double*>(z_depth_buffer.data()) : nullptr);
            }
            if (rc == ZR_OK && (use_reflection || use_refraction)) {
                // camera.hpp:490: either flag runs the second path and fills BOTH frames; the beauty image above is what the
                // reference's loop produces with the flags on (same stream prefix), so it is not rendered again
                const size_t npx = (size_t)image_width * image_height;
                reflection_buffer.assign(npx, color(0, 0, 0));
                refraction_buffer.assign(npx, color(0, 0, 0));
                rc = zr_render_passes(ctx, sc, &zc, &zenv, seed, nullptr, nullptr, reinterpret_cast<double*>(reflection_buffer.data()),
                                      reinterpret_cast<double*>(refraction_buffer.data()));
            }
        }
        if (rc != ZR_OK && rc != ZR_E_CANCELLED) std::cerr << "[zenith] render failed: " << zr_last_error() << "\n";
        if (sc) zr_scene_destroy(sc);
        }   // (the lease ends here: analyze() below takes its own)
        if (rc == ZR_OK && post.use_auto_exposure) {   // camera.hpp:258-266
            image_statistics stats = analyze(render_accumulator);
            post.last_stats = stats;
            post.exposure = static_cast<float>(post.apply_auto_exposure(stats));
        }
    }
};
