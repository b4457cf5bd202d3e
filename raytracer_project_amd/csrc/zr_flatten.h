// zr_flatten.h — from the caller's world list to the arrays the kernels read (included by zr_commit.cpp only): bounding boxes the way the reference's
// constructors compute them (Boxer), the classification of world entries (bare / baked / placed / wrapped), validation, and the Flattener, which turns the
// builder's tree into sibling-pair records, 4-wide quantised nodes and the primitive arrays in leaf order.
#pragma once
#include "zr_host_internal.h"

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

}  // namespace
