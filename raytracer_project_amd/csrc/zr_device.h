// zr_device.h — device-side data layout (HBM) and the per-ray arithmetic of the integrator, written for
// gfx950 (wave64).  FP64 throughout, as the reference (vec3 = 3 x double, /root/reference/vec3.hpp:7-115):
// every accept/reject decision of the reference is a double comparison and one flipped decision moves a
// pixel by ~1/spp >> the 1e-4 parity bound, so decisions are taken in FP64.  Only the BVH child boxes are
// FP32 (rounded outwards and padded): testing a superset of boxes cannot change the closest primitive.
//
// Each function names the reference code whose result it reproduces.  Nothing here is reference code:
// traversal is deferred-record (the BVH walk keeps only (t, leaf object) and the hit record of the winner
// is reconstructed once), which the reference's recursive virtual calls cannot do.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/zr_capi.h"
#include "../../include/zr_rng.h"

#include "zr_device_types.h"

namespace zr {

// ---- small vector algebra (same operation order as vec3.hpp) --------------------------------------
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(double t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); }
__device__ __forceinline__ V3 operator*(V3 v, double t) { return mk(t * v.x, t * v.y, t * v.z); }
__device__ __forceinline__ V3 vdiv(V3 v, double t) { double r = 1 / t; return mk(r * v.x, r * v.y, r * v.z); }  // vec3 operator/
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ double len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ double len(V3 a) { return sqrt(len2(a)); }
__device__ __forceinline__ V3 unit(V3 v) { double l = len(v); if (l < 1e-8) return mk(0, 0, 0); return vdiv(v, l); }
__device__ __forceinline__ double get(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
__device__ __forceinline__ V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (hi < v ? hi : v); }

struct Ray { V3 o, d; };
__device__ __forceinline__ V3 at(const Ray& r, double t) { return r.o + t * r.d; }

struct Rec {
    V3 p, n, tan, bit;
    double t, u, v;
    uint32_t mat;
    bool front;
};
__device__ __forceinline__ void set_face(Rec& rec, V3 rd, V3 outward) {  // hittable.hpp:22-25
    rec.front = dot(rd, outward) < 0;
    rec.n = rec.front ? outward : -outward;
}

struct Rng {
    uint64_t key, k;
    uint32_t bounce;
    // zr_stream_bits(key, k) = mix64(key + GOLDEN * (k + 1)); the product is one 64-bit multiply the counter can skip
    __device__ __forceinline__ double next() { k++; return zr_bits_to_unit(zr_mix64(key + ZR_GOLDEN * k)); }
    __device__ __forceinline__ double range(double a, double b) { return a + (b - a) * next(); }
};

// vec3.hpp:184-191.  (Round 4 tried letting the lanes that have their vector draw the pending lanes' next candidates — the wave runs this loop until its slowest
// lane accepts, ~7 iterations at 1.9 per lane — through LDS tables: bit-identical, and no faster on any workload: profiles/r4_experiments_ab.txt,
// scripts/dev/ruv_help_experiment.patch.  What did pay is ONE call site for all material kinds: scatter().)
__device__ inline V3 random_unit_vector(Rng& g) {
    for (;;) {
        double x = g.range(-1, 1);
        double y = g.range(-1, 1);
        double z = g.range(-1, 1);
        V3 p = mk(x, y, z);
        double l2 = len2(p);
        if (1e-160 < l2 && l2 <= 1) return vdiv(p, sqrt(l2));
    }
}

struct Counters { uint32_t nodes, sph, tri, cube, med; };

// ---- wrapper chain: ray into object space (translate/rotate_*/scale ::hit, first halves) -----------
__device__ inline void apply_op_ray(const zr_xform_op& op, Ray& r) {
    switch (op.kind) {
        case ZR_OP_TRANSLATE: r.o = r.o - mk(op.a[0], op.a[1], op.a[2]); break;
        case ZR_OP_ROTATE_Y: {
            double s = op.a[0], c = op.a[1];
            double ox = c * r.o.x + s * r.o.z, oz = -s * r.o.x + c * r.o.z;
            double dx = c * r.d.x + s * r.d.z, dz = -s * r.d.x + c * r.d.z;
            r.o.x = ox; r.o.z = oz; r.d.x = dx; r.d.z = dz;
        } break;
        case ZR_OP_ROTATE_X: {
            double s = op.a[0], c = op.a[1];
            double oy = c * r.o.y + s * r.o.z, oz = -s * r.o.y + c * r.o.z;
            double dy = c * r.d.y + s * r.d.z, dz = -s * r.d.y + c * r.d.z;
            r.o.y = oy; r.o.z = oz; r.d.y = dy; r.d.z = dz;
        } break;
        case ZR_OP_ROTATE_Z: {
            double s = op.a[0], c = op.a[1];
            double ox = c * r.o.x + s * r.o.y, oy = -s * r.o.x + c * r.o.y;
            double dx = c * r.d.x + s * r.d.y, dy = -s * r.d.x + c * r.d.y;
            r.o.x = ox; r.o.y = oy; r.d.x = dx; r.d.y = dy;
        } break;
        case ZR_OP_SCALE:
            r.o = mk(r.o.x / op.a[0], r.o.y / op.a[1], r.o.z / op.a[2]);
            r.d = mk(r.d.x / op.a[0], r.d.y / op.a[1], r.d.z / op.a[2]);
            break;
        default: break;  // ZR_OP_MATERIAL
    }
}
__device__ inline Ray chain_ray(const DScene& sc, uint32_t cf, uint32_t upto, Ray r) {
    for (uint32_t k = 0; k < upto; k++) apply_op_ray(sc.ops[cf + k], r);
    return r;
}
// second halves: hit record back to the wrapper's outer space.  rd_outer = direction of the ray the
// wrapper itself received.
__device__ inline void apply_op_rec(const zr_xform_op& op, V3 rd_outer, Rec& rec) {
    switch (op.kind) {
        case ZR_OP_TRANSLATE:  // translate.hpp:27-29
            rec.p = rec.p + mk(op.a[0], op.a[1], op.a[2]);
            set_face(rec, rd_outer, rec.n);
            break;
        case ZR_OP_ROTATE_Y: {  // rotate_y.hpp:58-70
            double s = op.a[0], c = op.a[1];
            V3 p = rec.p, n = rec.n;
            p.x = c * rec.p.x - s * rec.p.z; p.z = s * rec.p.x + c * rec.p.z;
            n.x = c * rec.n.x - s * rec.n.z; n.z = s * rec.n.x + c * rec.n.z;
            rec.p = p;
            set_face(rec, rd_outer, n);
        } break;
        case ZR_OP_ROTATE_X: {  // rotate_x.hpp:57-67: front_face not refreshed
            double s = op.a[0], c = op.a[1];
            V3 p = rec.p, n = rec.n;
            p.y = c * rec.p.y - s * rec.p.z; p.z = s * rec.p.y + c * rec.p.z;
            n.y = c * rec.n.y - s * rec.n.z; n.z = s * rec.n.y + c * rec.n.z;
            rec.p = p; rec.n = n;
        } break;
        case ZR_OP_ROTATE_Z: {  // rotate_z.hpp:54-64
            double s = op.a[0], c = op.a[1];
            V3 p = rec.p, n = rec.n;
            p.x = c * rec.p.x - s * rec.p.y; p.y = s * rec.p.x + c * rec.p.y;
            n.x = c * rec.n.x - s * rec.n.y; n.y = s * rec.n.x + c * rec.n.y;
            rec.p = p; rec.n = n;
        } break;
        case ZR_OP_SCALE:  // scale.hpp:29-33
            rec.p = mk(rec.p.x * op.a[0], rec.p.y * op.a[1], rec.p.z * op.a[2]);
            rec.n = unit(mk(rec.n.x / op.a[0], rec.n.y / op.a[1], rec.n.z / op.a[2]));
            break;
        default:  // material_instance.hpp:19-21
            rec.mat = op.mat;
            break;
    }
}

// ---- primitives: distance only (used while walking the tree) ---------------------------------------
// sphere.hpp:18-40.  surrounds(): strict on both ends.
__device__ __forceinline__ bool sphere_t(const double* s, const Ray& r, double tmin, double tmax, double& t) {
    V3 oc = mk(s[0], s[1], s[2]) - r.o;
    double a = len2(r.d);
    double h = dot(r.d, oc);
    double c = len2(oc) - s[3] * s[3];
    double disc = h * h - a * c;
    if (disc < 0) return false;
    double sq = sqrt(disc);
    double root = (h - sq) / a;
    if (!(tmin < root && tmax > root)) {
        root = (h + sq) / a;
        if (!(tmin < root && tmax > root)) return false;
    }
    t = root;
    return true;
}

// triangle.hpp:17-57, distance only.  Same decisions as the reference — degenerate (|N| < 1e-8), parallel
// (|N̂·d| < 1e-8), contains(t) inclusive, the three edge tests N·((v_{k+1}-v_k) x (p-v_k)) >= 0 — evaluated in the
// algebraically identical scaled-barycentric form, which needs no square root and a single division, and only
// for lanes that are inside:
//   with s = o - v0, q = d x e2, r = s x e1, det = e1·q = -(N·d):   N·C0 = |N|^2 v,  N·C2 = |N|^2 u,  N·C1 = |N|^2 (1-u-v)
//   u = (s·q)/det,  v = (d·r)/det,  t = (e2·r)/det = N·(v0 - o) / (N·d)
// The values differ from the reference's normalised-plane formula only in the last bits (~1e-15 relative).
__device__ __forceinline__ bool triangle_t(const double* v, const Ray& r, double tmin, double tmax, double& t) {
    V3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 n = cross(e1, e2);
    double nn = len2(n);
    if (nn < 1e-16) return false;                 // normal_length < 1e-8
    V3 q = cross(r.d, e2);
    double det = dot(e1, q);                      // = -(N·d)
    if (det * det < 1e-16 * nn) return false;     // |N̂·d| < 1e-8
    V3 s = r.o - v0;
    double un = dot(s, q);
    V3 rr = cross(s, e1);
    double vn = dot(r.d, rr);
    double sg = det < 0 ? -1.0 : 1.0;
    double ad = det * sg;
    un *= sg; vn *= sg;
    if (un < 0 || vn < 0 || un + vn > ad) return false;
    double tt = dot(e2, rr) / det;
    if (!(tmin <= tt && tt <= tmax)) return false;
    t = tt;
    return true;
}

// cube.hpp:44-73 (slabs about the ORIGIN, not the centre: cube.hpp:57-58)
__device__ __forceinline__ bool cube_t(const double* q, const Ray& r, double tmin, double tmax, double& t) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double he = q[i];
        double inv = 1.0 / get(r.d, i);
        double o = get(r.o, i);
        double t0 = (-he - o) * inv;
        double t1 = (he - o) * inv;
        if (inv < 0.0) { double x = t0; t0 = t1; t1 = x; }
        tmin = fmax(t0, tmin);
        tmax = fmin(t1, tmax);
        if (tmax < tmin) return false;
    }
    t = tmin;
    return true;
}

// cube_t with 1 / d given: the caller computed it once for a ray that meets several cubes (the fused small-scene kernel: a Cornell box is seven of them, and a
// translate leaves the direction alone, translate.hpp:15-22) — the same division on the same operand, so the same bits, 3 instead of 3 per cube
__device__ __forceinline__ bool cube_t_inv(const double* q, const Ray& r, V3 inv_d, double tmin, double tmax, double& t) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double he = q[i];
        double inv = get(inv_d, i);
        double o = get(r.o, i);
        double t0 = (-he - o) * inv;
        double t1 = (he - o) * inv;
        if (inv < 0.0) { double x = t0; t0 = t1; t1 = x; }
        tmin = fmax(t0, tmin);
        tmax = fmin(t1, tmax);
        if (tmax < tmin) return false;
    }
    t = tmin;
    return true;
}

// A placed cube: the ray goes through translate::hit's, rotate_y::hit's and scale::hit's first halves (translate.hpp:15-22, rotate_y.hpp:44-56,
// scale.hpp:20-24) exactly as chain_ray would apply them, then cube::hit — the wrapper parameters come with the record instead of from an op list
// SCALED = false: the caller knows that no record carries a scale (the fused small-scene kernel: the host sends worlds with scaled placed cubes
// through the pipeline instead), and the division code is not compiled into it
template <bool SCALED = true>
__device__ __forceinline__ Ray pcube_ray(const double* q, Ray r) {
    r.o = r.o - mk(q[6], q[7], q[8]);
    if (q[11] != 0.0) {
        const double s = q[9], c = q[10];
        const double ox = c * r.o.x + s * r.o.z, oz = -s * r.o.x + c * r.o.z;
        const double dx = c * r.d.x + s * r.d.z, dz = -s * r.d.x + c * r.d.z;
        r.o.x = ox; r.o.z = oz; r.d.x = dx; r.d.z = dz;
    }
    if (SCALED && q[15] != 0.0) {   // scale::hit's first half (scale.hpp:20-24), the arithmetic of apply_op_ray
        r.o = mk(r.o.x / q[12], r.o.y / q[13], r.o.z / q[14]);
        r.d = mk(r.d.x / q[12], r.d.y / q[13], r.d.z / q[14]);
    }
    return r;
}
template <bool SCALED = true>
__device__ __forceinline__ bool pcube_t(const double* q, const Ray& r, double tmin, double tmax, double& t) {
    const Ray lr = pcube_ray<SCALED>(q, r);
    return cube_t(q, lr, tmin, tmax, t);
}
// ... with the world ray's 1 / d given: used as it is when the record neither rotates nor scales (the direction is the world ray's)
template <bool SCALED = true>
__device__ __forceinline__ bool pcube_t_inv(const double* q, const Ray& r, V3 inv_d, double tmin, double tmax, double& t) {
    const Ray lr = pcube_ray<SCALED>(q, r);
    if (q[11] != 0.0 || (SCALED && q[15] != 0.0)) return cube_t(q, lr, tmin, tmax, t);
    return cube_t_inv(q, lr, inv_d, tmin, tmax, t);
}

__device__ inline bool bare_t(const DScene& sc, uint32_t kind, uint32_t idx, const Ray& r, double tmin, double tmax, double& t) {
    if (kind == ZR_PRIM_SPHERE) return sphere_t(sc.spheres + (size_t)idx * 4, r, tmin, tmax, t);
    if (kind == ZR_PRIM_CUBE) return cube_t(sc.cubes + (size_t)idx * 6, r, tmin, tmax, t);
    return triangle_t(sc.tri_v + (size_t)idx * ZR_TRI_STRIDE, r, tmin, tmax, t);
}

// constant_medium.hpp:39-77, distance only
__device__ inline bool medium_t(const DScene& sc, uint32_t idx, const Ray& r, double tmin, double tmax, const Rng& g, double& t) {
    const DMedium& m = sc.media[idx];
    Ray br = chain_ray(sc, m.chain_first, m.chain_count, r);
    double t1, t2;
    const double inf = __builtin_huge_val();
    if (!bare_t(sc, m.btype, m.bindex, br, -inf, inf, t1)) return false;
    if (!bare_t(sc, m.btype, m.bindex, br, t1 + 0.0001, inf, t2)) return false;
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < 0) t1 = 0;
    double rl = len(r.d);
    double inside = (t2 - t1) * rl;
    double xi = zr_bits_to_unit(zr_medium_bits(g.key, g.bounce, m.id));
    double hd = m.neg_inv_density * log(xi);
    if (hd > inside) return false;
    t = t1 + hd / rl;
    return true;
}

// a medium whose boundary is an unwrapped sphere or cube (DMedium::chain_count == 0): medium_t without the op-list loop and the
// triangle branch — the same arithmetic, since an empty chain leaves the ray as it is
__device__ inline bool medium_plain_t(const DScene& sc, uint32_t idx, const Ray& r, double tmin, double tmax, const Rng& g, double& t) {
    const DMedium& m = sc.media[idx];
    double t1, t2;
    const double inf = __builtin_huge_val();
    if (m.btype == ZR_PRIM_SPHERE) {
        const double* q = sc.spheres + (size_t)m.bindex * 4;
        if (!sphere_t(q, r, -inf, inf, t1)) return false;
        if (!sphere_t(q, r, t1 + 0.0001, inf, t2)) return false;
    } else {
        const double* q = sc.cubes + (size_t)m.bindex * 6;
        if (!cube_t(q, r, -inf, inf, t1)) return false;
        if (!cube_t(q, r, t1 + 0.0001, inf, t2)) return false;
    }
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < 0) t1 = 0;
    double rl = len(r.d);
    double inside = (t2 - t1) * rl;
    double xi = zr_bits_to_unit(zr_medium_bits(g.key, g.bounce, m.id));
    double hd = m.neg_inv_density * log(xi);
    if (hd > inside) return false;
    t = t1 + hd / rl;
    return true;
}

// any leaf object
__device__ inline bool object_t(const DScene& sc, uint32_t kind, uint32_t idx, const Ray& r, double tmin, double tmax, const Rng& g,
                                double& t) {
    if (kind == ZR_KIND_WRAPPED) {
        const DWrapped w = sc.wrapped[idx];
        Ray lr = chain_ray(sc, w.chain_first, w.chain_count, r);
        if (w.type == ZR_PRIM_MEDIUM) return medium_t(sc, w.index, lr, tmin, tmax, g, t);
        return bare_t(sc, w.type, w.index, lr, tmin, tmax, t);
    }
    if (kind == ZR_KIND_PCUBE) return pcube_t(sc.pcubes + (size_t)idx * ZR_PCUBE_STRIDE, r, tmin, tmax, t);
    if (kind == ZR_PRIM_MEDIUM) return medium_t(sc, idx, r, tmin, tmax, g, t);
    return bare_t(sc, kind, idx, r, tmin, tmax, t);
}

__device__ __forceinline__ void slab_constants(double d, double o, double& id, double& o_id);
// ---- two-level BVH: a run of triangles placed under a wrapper chain (ZR_KIND_INSTANCE) -----------------------------------
// closest triangle of a run's subtree (sibling-pair records whose leaves are triangles) for a ray already mapped into the
// run's space; the walk of closest_hit below, on a stack of its own.  n_boxes / n_tris: tests made (instrumented builds).
__device__ inline bool run_hit(const DScene& sc, uint32_t root, const Ray& r, double tmin, double tmax, double& t_out, uint32_t& tri_out,
                               uint32_t& n_boxes, uint32_t& n_tris) {
    double idx_, idy_, idz_, ox_, oy_, oz_;
    slab_constants(r.d.x, r.o.x, idx_, ox_); slab_constants(r.d.y, r.o.y, idy_, oy_); slab_constants(r.d.z, r.o.z, idz_, oz_);
    double tbest = tmax;
    bool found = false;
    uint32_t stk[ZR_STACK_DEPTH];
    int sp = 0;
    uint32_t cur = root;
    for (;;) {
        const NodePair* np = sc.nodes + cur;
        const float4 q0 = reinterpret_cast<const float4*>(np)[0];
        const float4 q1 = reinterpret_cast<const float4*>(np)[1];
        const float4 q2 = reinterpret_cast<const float4*>(np)[2];
        const uint4 q3 = reinterpret_cast<const uint4*>(np)[3];
        n_boxes += 2;
        double tn[2], tf[2];
        {
            double a0 = fma((double)q0.x, idx_, -ox_), a1 = fma((double)q1.z, idx_, -ox_);
            double b0 = fma((double)q0.y, idy_, -oy_), b1 = fma((double)q1.w, idy_, -oy_);
            double c0 = fma((double)q0.z, idz_, -oz_), c1 = fma((double)q2.x, idz_, -oz_);
            tn[0] = fmax(fmax(fmin(a0, a1), fmin(b0, b1)), fmax(fmin(c0, c1), tmin));
            tf[0] = fmin(fmin(fmax(a0, a1), fmax(b0, b1)), fmin(fmax(c0, c1), tbest));
        }
        {
            double a0 = fma((double)q0.w, idx_, -ox_), a1 = fma((double)q2.y, idx_, -ox_);
            double b0 = fma((double)q1.x, idy_, -oy_), b1 = fma((double)q2.z, idy_, -oy_);
            double c0 = fma((double)q1.y, idz_, -oz_), c1 = fma((double)q2.w, idz_, -oz_);
            tn[1] = fmax(fmax(fmin(a0, a1), fmin(b0, b1)), fmax(fmin(c0, c1), tmin));
            tf[1] = fmin(fmin(fmax(a0, a1), fmax(b0, b1)), fmin(fmax(c0, c1), tbest));
        }
        const bool h[2] = {tn[0] <= tf[0], tn[1] <= tf[1]};
        const uint32_t c[2] = {q3.x, q3.y};
        const uint32_t m[2] = {q3.z, q3.w};
        uint32_t next = 0xFFFFFFFFu, defer = 0xFFFFFFFFu;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (!h[s]) continue;
            if (m[s] != 0) {
                const uint32_t cnt = m[s] & 0xFFFFu;
                for (uint32_t k = 0; k < cnt; k++) {
                    double t;
                    n_tris++;
                    if (triangle_t(sc.tri_v + (size_t)(c[s] + k) * ZR_TRI_STRIDE, r, tmin, tbest, t)) { tbest = t; tri_out = c[s] + k; found = true; }
                }
            } else {
                if (next == 0xFFFFFFFFu) next = c[s]; else defer = c[s];
            }
        }
        if (defer != 0xFFFFFFFFu) {
            if (tn[1] < tn[0]) { uint32_t x = next; next = defer; defer = x; }
            if (sp < ZR_STACK_DEPTH) stk[sp++] = defer;
        }
        if (next == 0xFFFFFFFFu) {
            if (sp == 0) break;
            next = stk[--sp];
        }
        cur = next;
    }
    t_out = tbest;
    return found;
}
// the instance's test: the wrappers' ray maps outermost first (what translate / rotate_* / scale ::hit do before calling the
// child), then the child's own tree; t is the same in both spaces
__device__ inline bool instance_t(const DScene& sc, uint32_t inst, const Ray& r, double tmin, double tmax, double& t, uint32_t& tri,
                                  uint32_t& n_boxes, uint32_t& n_tris) {
    const DInstance in = sc.insts[inst];
    const Ray lr = chain_ray(sc, in.chain_first, in.chain_count, r);
    return run_hit(sc, in.root, lr, tmin, tmax, t, tri, n_boxes, n_tris);
}

// material id a hit on leaf object (kind, idx) will carry: the primitive's own, or that of the outermost material_instance
// of its wrapper chain (material_instance.hpp:19-21; object_rec applies the chain inside-out, so the outermost one wins)
__device__ inline uint32_t object_material(const DScene& sc, uint32_t kind, uint32_t idx) {
    uint32_t type = kind, index = idx, mat_override = 0xFFFFFFFEu;
    if (kind == ZR_KIND_WRAPPED || (kind & 0xFFu) == ZR_KIND_INSTANCE) {
        DWrapped w;
        if (kind == ZR_KIND_WRAPPED) w = sc.wrapped[idx];
        else { const DInstance in = sc.insts[kind >> 8]; w.type = ZR_PRIM_TRIANGLE; w.index = idx; w.chain_first = in.chain_first; w.chain_count = in.chain_count; }
        type = w.type; index = w.index;
        for (uint32_t k = 0; k < w.chain_count; k++)
            if (sc.ops[w.chain_first + k].kind == ZR_OP_MATERIAL) { mat_override = sc.ops[w.chain_first + k].mat; break; }
    }
    if (mat_override != 0xFFFFFFFEu) return mat_override;
    if (type == ZR_KIND_PCUBE) return sc.pcube_mat[index];
    if (type == ZR_PRIM_SPHERE) { const uint32_t m = sc.sphere_mat[index]; return m == 0xFFFFFFFFu ? m : (m & 0x7FFFFFFFu); }
    if (type == ZR_PRIM_TRIANGLE) return (uint32_t)__double_as_longlong(sc.tri_s[(size_t)index * 20 + 18]);
    if (type == ZR_PRIM_CUBE) return sc.cube_mat[index];
    return sc.media[index].mat;
}

// ---- primitives: full hit record of the winner ------------------------------------------------------
// `full` = false: u, v, tangent, bitangent are only computed when the hit's material reads them (a texture that
// depends on u/v, or a bump map: zr_material::pad_ set by the host).  They are pure functions of the hit, so
// skipping them when nothing consumes them cannot change any result; the reference computes acos/atan2 on every
// accepted candidate (sphere.hpp:47,70-79).
__device__ __forceinline__ bool mat_needs_uv(const DScene& sc, uint32_t mat) { return mat < sc.n_mats && sc.mats[mat].pad_ != 0; }

__device__ inline void sphere_rec(const DScene& sc, uint32_t idx, const Ray& r, double t, Rec& rec, bool full) {  // sphere.hpp:42-79
    const double* s = sc.spheres + (size_t)idx * 4;
    V3 center = mk(s[0], s[1], s[2]);
    rec.t = t;
    rec.p = at(r, t);
    V3 outward = vdiv(rec.p - center, s[3]);
    set_face(rec, r.d, outward);
    rec.mat = sc.sphere_mat[idx];
    if (rec.mat != 0xFFFFFFFFu && (rec.mat & 0x80000000u)) { rec.mat &= 0x7FFFFFFFu; rec.front = true; }   // baked from under a translate (zr_flatten.h)
    if (!full && !mat_needs_uv(sc, rec.mat)) { rec.u = 0; rec.v = 0; rec.tan = mk(0, 0, 0); rec.bit = mk(0, 0, 0); return; }
    double theta = acos(-outward.y);
    double phi = atan2(-outward.z, outward.x) + 3.14159265358979323846;
    rec.u = phi / (2 * 3.14159265358979323846);
    rec.v = theta / 3.14159265358979323846;
    rec.tan = cross(mk(0, 1, 0), rec.n);
    if (len2(rec.tan) < 0.001) rec.tan = cross(mk(0, 0, 1), rec.n);
    rec.tan = unit(rec.tan);
    rec.bit = cross(rec.n, rec.tan);
}

__device__ inline void triangle_rec(const DScene& sc, uint32_t idx, const Ray& r, double t, Rec& rec) {  // triangle.hpp:40-79
    const double* v = sc.tri_s + (size_t)idx * 20;  // one record = everything the hit record needs
    const double* nn = v + 9;
    V3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
    V3 normal = cross(v1 - v0, v2 - v0);
    V3 p = at(r, t);
    V3 C0 = cross(v1 - v0, p - v0);
    V3 C2 = cross(v0 - v2, p - v2);
    double area2 = dot(normal, normal);
    double u = dot(normal, C2) / area2;
    double w_ = dot(normal, C0) / area2;
    double w0 = 1.0 - u - w_;
    V3 smooth = unit(w0 * ld3(nn) + u * ld3(nn + 3) + w_ * ld3(nn + 6));
    rec.t = t;
    rec.p = p;
    rec.mat = (uint32_t)__double_as_longlong(v[18]);
    set_face(rec, r.d, smooth);
    if ((uint32_t)__double_as_longlong(v[19]) & 1u) rec.front = true;   // baked from under a translate / rotate_y (zr_flatten.h)
    // u, v, tangent, bitangent are not written by triangle::hit: fresh-record values
    rec.u = 0; rec.v = 0; rec.tan = mk(0, 0, 0); rec.bit = mk(0, 0, 0);
}

// uv = false: u, v, tangent and bitangent are left zero — nothing reads them (the hit's material looks up no image and has no bump map: mat_needs_uv);
// the face's u, v cost two divisions per hit, and a Cornell box is all cubes
__device__ inline void cube_rec_q(const double* q, uint32_t mat, const Ray& r, double t, Rec& rec, bool uv = true) {  // cube.hpp:73-142
    V3 he = ld3(q), center = ld3(q + 3);
    rec.t = t;
    rec.p = at(r, t);
    V3 p = rec.p - center;
    const double EPS = 1e-3;
    if (!uv) {
        if (fabs(p.x + he.x) < EPS) rec.n = mk(-1, 0, 0);
        else if (fabs(p.x - he.x) < EPS) rec.n = mk(1, 0, 0);
        else if (fabs(p.y + he.y) < EPS) rec.n = mk(0, -1, 0);
        else if (fabs(p.y - he.y) < EPS) rec.n = mk(0, 1, 0);
        else if (fabs(p.z + he.z) < EPS) rec.n = mk(0, 0, -1);
        else rec.n = mk(0, 0, 1);
        rec.u = 0; rec.v = 0; rec.tan = mk(0, 0, 0); rec.bit = mk(0, 0, 0);
        rec.mat = mat;
        set_face(rec, r.d, rec.n);
        return;
    }
    if (fabs(p.x + he.x) < EPS) {
        rec.n = mk(-1, 0, 0); rec.u = (p.z + he.z) / (2 * he.z); rec.v = (p.y + he.y) / (2 * he.y); rec.tan = mk(0, 0, 1);
    } else if (fabs(p.x - he.x) < EPS) {
        rec.n = mk(1, 0, 0); rec.u = (p.z + he.z) / (2 * he.z); rec.v = (p.y + he.y) / (2 * he.y); rec.tan = mk(0, 0, -1);
    } else if (fabs(p.y + he.y) < EPS) {
        rec.n = mk(0, -1, 0); rec.u = (p.x + he.x) / (2 * he.x); rec.v = (p.z + he.z) / (2 * he.z); rec.tan = mk(1, 0, 0);
    } else if (fabs(p.y - he.y) < EPS) {
        rec.n = mk(0, 1, 0); rec.u = (p.x + he.x) / (2 * he.x); rec.v = (p.z + he.z) / (2 * he.z); rec.tan = mk(-1, 0, 0);
    } else if (fabs(p.z + he.z) < EPS) {
        rec.n = mk(0, 0, -1); rec.u = (he.x - p.x) / (2 * he.x); rec.v = (p.y + he.y) / (2 * he.y); rec.tan = mk(-1, 0, 0);
    } else {
        rec.n = mk(0, 0, 1); rec.u = (p.x + he.x) / (2 * he.x); rec.v = (p.y + he.y) / (2 * he.y); rec.tan = mk(1, 0, 0);
    }
    rec.bit = cross(rec.n, rec.tan);
    rec.mat = mat;
    set_face(rec, r.d, rec.n);
}
__device__ inline void cube_rec(const DScene& sc, uint32_t idx, const Ray& r, double t, Rec& rec, bool full = true) {
    const uint32_t mat = sc.cube_mat[idx];
    cube_rec_q(sc.cubes + (size_t)idx * 6, mat, r, t, rec, full || mat_needs_uv(sc, mat));
}

__device__ inline void bare_rec(const DScene& sc, uint32_t kind, uint32_t idx, const Ray& r, double t, Rec& rec, bool full) {
    if (kind == ZR_PRIM_SPHERE) sphere_rec(sc, idx, r, t, rec, full);
    else if (kind == ZR_PRIM_CUBE) cube_rec(sc, idx, r, t, rec, full);
    else if (kind == ZR_PRIM_TRIANGLE) triangle_rec(sc, idx, r, t, rec);
    else {  // medium: constant_medium.hpp:70-75
        rec.t = t;
        rec.p = at(r, t);
        rec.n = mk(1, 0, 0);
        rec.front = true;
        rec.mat = sc.media[idx].mat;
        rec.u = 0; rec.v = 0; rec.tan = mk(0, 0, 0); rec.bit = mk(0, 0, 0);
    }
}

// hit record of leaf object (kind, idx) hit by world ray r at distance t
// `full`: compute every field (known-answer entry); otherwise u/v/tangent only when the material reads them
template <bool PSCALE = true>
__device__ inline void object_rec(const DScene& sc, uint32_t kind, uint32_t idx, const Ray& r, double t, Rec& rec, bool full = false) {
    if (kind == ZR_KIND_PCUBE) {   // cube::hit's record in object space, then the second halves of scale::hit, rotate_y::hit and translate::hit, inside-out
        const double* q = sc.pcubes + (size_t)idx * ZR_PCUBE_STRIDE;
        const Ray lr = pcube_ray<PSCALE>(q, r);
        cube_rec_q(q, sc.pcube_mat[idx], lr, t, rec, full || mat_needs_uv(sc, sc.pcube_mat[idx]));
        zr_xform_op op; op.mat = 0;
        if (PSCALE && q[15] != 0.0) { op.kind = ZR_OP_SCALE; op.a[0] = q[12]; op.a[1] = q[13]; op.a[2] = q[14]; apply_op_rec(op, r.d, rec); }   // scale.hpp:29-33 (reads no ray)
        if (q[11] != 0.0) { op.kind = ZR_OP_ROTATE_Y; op.a[0] = q[9]; op.a[1] = q[10]; op.a[2] = 0; apply_op_rec(op, r.d, rec); }   // the ray rotate_y received: translated only, same direction
        op.kind = ZR_OP_TRANSLATE; op.a[0] = q[6]; op.a[1] = q[7]; op.a[2] = q[8];
        apply_op_rec(op, r.d, rec);
        return;
    }
    const bool placed_run = (kind & 0xFFu) == ZR_KIND_INSTANCE;   // a triangle of a run: the record of a triangle under the instance's chain
    if (kind != ZR_KIND_WRAPPED && !placed_run) { bare_rec(sc, kind, idx, r, t, rec, full); return; }
    DWrapped w;
    if (placed_run) { const DInstance in = sc.insts[kind >> 8]; w.type = ZR_PRIM_TRIANGLE; w.index = idx; w.chain_first = in.chain_first; w.chain_count = in.chain_count; }
    else w = sc.wrapped[idx];
    Ray lr = chain_ray(sc, w.chain_first, w.chain_count, r);
    bare_rec(sc, w.type, w.index, lr, t, rec, true);  // a material_instance in the chain may replace the material
    for (int k = (int)w.chain_count - 1; k >= 0; k--) {
        Ray outer = chain_ray(sc, w.chain_first, (uint32_t)k, r);
        apply_op_rec(sc.ops[w.chain_first + k], outer.d, rec);
    }
}

// Slab constants of one axis for t = fma(P, id, -o_id).  With a zero direction component id = +-inf, and P * inf - o * inf
// is -inf for one plane and NaN for the other whenever o lies between them: fmin/fmax would then keep the -inf and
// cull a box the ray runs through.  Such an axis (and one whose o / d overflows) gets id = 0, o_id = NaN instead:
// both planes evaluate to NaN, fmin/fmax ignore them and the slab is dropped — conservative.  (aabb.hpp:44-66 relies on
// IEEE inf arithmetic of (P - o) * inf, which has no such mixed case.)
__device__ __forceinline__ void slab_constants(double d, double o, double& id, double& o_id) {
    id = 1.0 / d;
    o_id = o * id;
    if (!(fabs(id) < 1e300) || !(fabs(o_id) < 1e300)) { id = 0.0; o_id = __builtin_nan(""); }
}

// ---- closest hit: BVH walk (replaces bvh_node::hit, bvh.hpp:46-54,112-118 + aabb::hit, aabb.hpp:44-66) ----
// `stack` is this lane's column of the workgroup's LDS traversal stack: entry i lives at stack[i * stride].
// Returns the leaf object and distance of the closest hit in the interval (tmin, tmax) — each primitive applies its own
// reading of the bounds, as in the reference (sphere: open, triangle: closed, cube: clamps to them).
template <bool COUNT>
__device__ inline bool closest_hit(const DScene& sc, const Ray& r, double tmin, const Rng& g, uint32_t* stack, int stride,
                                   double& t_out, uint32_t& kind_out, uint32_t& idx_out, Counters& ctr, double tmax = __builtin_huge_val()) {
    double idx_, idy_, idz_, ox_, oy_, oz_;
    slab_constants(r.d.x, r.o.x, idx_, ox_); slab_constants(r.d.y, r.o.y, idy_, oy_); slab_constants(r.d.z, r.o.z, idz_, oz_);
    double tbest = tmax;
    uint32_t kbest = 0xFFFFFFFFu, ibest = 0;
    int sp = 0;
    uint32_t cur = 0;  // NodePair index
    for (;;) {
        const NodePair* np = sc.nodes + cur;
        // one 64-B record: 4 x 16-B loads
        const float4 q0 = reinterpret_cast<const float4*>(np)[0];
        const float4 q1 = reinterpret_cast<const float4*>(np)[1];
        const float4 q2 = reinterpret_cast<const float4*>(np)[2];
        const uint4 q3 = reinterpret_cast<const uint4*>(np)[3];
        // lo[0] = q0.xyz, lo[1] = q0.w q1.xy, hi[0] = q1.zw q2.x, hi[1] = q2.yzw
        if (COUNT) ctr.nodes += 2;
        double tn[2], tf[2];
        {
            double a0 = fma((double)q0.x, idx_, -ox_), a1 = fma((double)q1.z, idx_, -ox_);
            double b0 = fma((double)q0.y, idy_, -oy_), b1 = fma((double)q1.w, idy_, -oy_);
            double c0 = fma((double)q0.z, idz_, -oz_), c1 = fma((double)q2.x, idz_, -oz_);
            tn[0] = fmax(fmax(fmin(a0, a1), fmin(b0, b1)), fmax(fmin(c0, c1), tmin));
            tf[0] = fmin(fmin(fmax(a0, a1), fmax(b0, b1)), fmin(fmax(c0, c1), tbest));
        }
        {
            double a0 = fma((double)q0.w, idx_, -ox_), a1 = fma((double)q2.y, idx_, -ox_);
            double b0 = fma((double)q1.x, idy_, -oy_), b1 = fma((double)q2.z, idy_, -oy_);
            double c0 = fma((double)q1.y, idz_, -oz_), c1 = fma((double)q2.w, idz_, -oz_);
            tn[1] = fmax(fmax(fmin(a0, a1), fmin(b0, b1)), fmax(fmin(c0, c1), tmin));
            tf[1] = fmin(fmin(fmax(a0, a1), fmax(b0, b1)), fmin(fmax(c0, c1), tbest));
        }
        // an axis the ray does not move along evaluates to NaN on both planes (slab_constants): fmin/fmax ignore it
        bool h0 = tn[0] <= tf[0], h1 = tn[1] <= tf[1];
        const uint32_t c[2] = {q3.x, q3.y};
        const uint32_t m[2] = {q3.z, q3.w};
        // leaves first: they can only shrink tbest
        uint32_t next = 0xFFFFFFFFu, defer = 0xFFFFFFFFu;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            bool h = s == 0 ? h0 : h1;
            if (!h) continue;
            if (m[s] != 0) {
                uint32_t kind = (m[s] >> 16) - 1, cnt = m[s] & 0xFFFFu;
                for (uint32_t k = 0; k < cnt; k++) {
                    double t;
                    if (COUNT) {
                        uint32_t kk = kind;
                        if (kk == ZR_KIND_WRAPPED) kk = sc.wrapped[c[s] + k].type;
                        if (kk == ZR_PRIM_SPHERE) ctr.sph++; else if (kk == ZR_PRIM_TRIANGLE) ctr.tri++; else if (kk == ZR_PRIM_CUBE || kk == ZR_KIND_PCUBE) ctr.cube++; else ctr.med++;
                    }
                    if (kind == ZR_KIND_INSTANCE) {
                        uint32_t tri = 0, nb = 0, nt = 0;
                        if (instance_t(sc, c[s] + k, r, tmin, tbest, t, tri, nb, nt)) { tbest = t; kbest = ZR_KIND_INSTANCE | ((c[s] + k) << 8); ibest = tri; }
                        if (COUNT) { ctr.nodes += nb; ctr.tri += nt; ctr.med--; }   // (the line above counted the instance itself as "other")
                    } else
                    if (object_t(sc, kind, c[s] + k, r, tmin, tbest, g, t)) { tbest = t; kbest = kind; ibest = c[s] + k; }
                }
            } else {
                if (next == 0xFFFFFFFFu) next = c[s]; else defer = c[s];
            }
        }
        if (defer != 0xFFFFFFFFu) {
            // both children are internal and hit: descend into the nearer one first
            if (tn[1] < tn[0]) { uint32_t x = next; next = defer; defer = x; }
            stack[sp * stride] = defer; sp++;
        }
        if (next == 0xFFFFFFFFu) {
            if (sp == 0) break;
            sp--; next = stack[sp * stride];
        }
        cur = next;
    }
    t_out = tbest; kind_out = kbest; idx_out = ibest;
    return kbest != 0xFFFFFFFFu;
}

// ---- textures (texture.hpp:50-78, 96-98, 118-126) -----------------------------------------------------
__device__ inline V3 tex_value(const DScene& sc, uint32_t id, double u, double v, V3 p) {
    for (int guard = 0; guard < 16; guard++) {
        const zr_texture& t = sc.texs[id];
        if (t.kind == ZR_TEX_SOLID) return mk(t.color[0], t.color[1], t.color[2]);
        if (t.kind == ZR_TEX_CHECKER) {
            int xi = (int)floor(t.inv_scale * p.x);
            int yi = (int)floor(t.inv_scale * p.y);
            int zi = (int)floor(t.inv_scale * p.z);
            bool even = (xi + yi + zi) % 2 == 0;
            id = even ? t.even : t.odd;
            continue;
        }
        if (t.width == 0 || t.height == 0) return mk(0.0, 1.0, 1.0);
        int width = (int)t.width, height = (int)t.height;
        u = u - floor(u);
        int i = (int)(u * width);
        int j = (int)(v * height);
        i = i < 0 ? 0 : (i > width - 1 ? width - 1 : i);
        j = j < 0 ? 0 : (j > height - 1 ? height - 1 : j);
        const unsigned char* base = sc.texels + t.texel_offset;
        size_t o = ((size_t)j * width + i) * 3;
        if (t.kind == ZR_TEX_IMAGE_F32) {
            const float* px = (const float*)base + o;
            return mk((double)px[0], (double)px[1], (double)px[2]);
        }
        const double s = 1.0 / 255.0;
        const unsigned char* px = base + o;
        return mk(s * px[0], s * px[1], s * px[2]);
    }
    return mk(0, 0, 0);
}

__device__ inline V3 bumped_normal(const DScene& sc, const Rec& rec, uint32_t bump, double strength) {  // material.hpp:35-54
    double du = 1.0 / 1024.0, dv = 1.0 / 1024.0;
    double hc = tex_value(sc, bump, rec.u, rec.v, rec.p).x;
    double hu = tex_value(sc, bump, rec.u + du, rec.v, rec.p).x;
    double hv = tex_value(sc, bump, rec.u, rec.v + dv, rec.p).x;
    double fu = (hu - hc) * strength;
    double fv = (hv - hc) * strength;
    return unit(rec.n - (fu * rec.tan) - (fv * rec.bit));
}

// ---- materials --------------------------------------------------------------------------------------
__device__ inline V3 reflect(V3 v, V3 n) { return v - (2 * dot(v, n)) * n; }

__device__ inline V3 emitted(const DScene& sc, const Rec& rec) {  // material.hpp:12-14, 261-263
    if (rec.mat >= sc.n_mats) return mk(0, 0, 0);
    const zr_material& m = sc.mats[rec.mat];
    if (m.kind == ZR_MAT_LIGHT) return tex_value(sc, m.tex, rec.u, rec.v, rec.p);
    return mk(0, 0, 0);
}

// material::get_albedo (material.hpp:29-31, 99-102, 154-156, 226-229, 266-275)
__device__ inline V3 get_albedo(const DScene& sc, const Rec& rec) {
    if (rec.mat >= sc.n_mats) return mk(0, 0, 0);
    const zr_material& m = sc.mats[rec.mat];
    switch (m.kind) {
        case ZR_MAT_LAMBERTIAN:
        case ZR_MAT_METAL: return tex_value(sc, m.tex, rec.u, rec.v, rec.p);
        case ZR_MAT_DIELECTRIC: return mk(1.0, 1.0, 1.0);
        case ZR_MAT_LIGHT: { V3 c = tex_value(sc, m.tex, rec.u, rec.v, rec.p); return mk(fmin(c.x, 1.0), fmin(c.y, 1.0), fmin(c.z, 1.0)); }
        default: return mk(0, 0, 0);  // isovolumetric keeps material's default
    }
}

// returns false when the path ends (absorbed / light).  A null material (UB in the reference) absorbs.
__device__ inline bool scatter(const DScene& sc, const Ray& rin, const Rec& rec, V3& att, Ray& out, Rng& g) {
    if (rec.mat >= sc.n_mats) return false;
    const zr_material& m = sc.mats[rec.mat];
    // lambertian, metal and the isotropic phase function each begin their draws with random_unit_vector() (material.hpp:82,141, constant_medium.hpp:15; bump maps draw
    // nothing): taken HERE, once, for the lanes of all three kinds together — in a wave that mixes them (the fused kernel: walls, fog) the rejection sampler, which
    // runs until its slowest lane accepts, used to run once per kind
    // ... and the same for their albedo texture and bump map: one look-up site instead of three (neither draws)
    V3 ruv = mk(0, 0, 0), tv = mk(0, 0, 0), wn = rec.n;
    if (m.kind == ZR_MAT_LAMBERTIAN || m.kind == ZR_MAT_METAL || m.kind == ZR_MAT_ISOTROPIC) {
        ruv = random_unit_vector(g);
        tv = tex_value(sc, m.tex, rec.u, rec.v, rec.p);
    }
    if (m.kind <= ZR_MAT_DIELECTRIC && m.bump_tex != ZR_NO_TEXTURE) wn = bumped_normal(sc, rec, m.bump_tex, m.bump_strength);   // lambertian, metal, dielectric
    V3 ud = mk(0, 0, 0);   // unit_vector(r_in.direction()): metal and dielectric (material.hpp:139,199)
    if (m.kind == ZR_MAT_METAL || m.kind == ZR_MAT_DIELECTRIC) ud = unit(rin.d);
    switch (m.kind) {
        case ZR_MAT_LAMBERTIAN: {  // material.hpp:74-96
            V3 dir = wn + ruv;
            if (fabs(dir.x) < 1e-8 && fabs(dir.y) < 1e-8 && fabs(dir.z) < 1e-8) dir = wn;
            out.o = rec.p + (rec.n * 0.0001);
            out.d = dir;
            att = tv;
            return true;
        }
        case ZR_MAT_METAL: {  // material.hpp:129-151
            V3 refl = reflect(ud, wn);
            V3 dir = unit(refl + (m.param * ruv));
            out.o = rec.p + (0.0001 * rec.n);
            out.d = dir;
            att = tv;
            return dot(dir, rec.n) > 0;
        }
        case ZR_MAT_DIELECTRIC: {  // material.hpp:192-224, 237-241
            att = mk(m.tint[0], m.tint[1], m.tint[2]);
            double ri = rec.front ? (1.0 / m.param) : m.param;
            double ct = fmin(dot(-ud, wn), 1.0);
            double st = sqrt(1.0 - ct * ct);
            bool refl = ri * st > 1.0;
            if (!refl) {
                double r0 = (1 - ri) / (1 + ri);
                r0 = r0 * r0;
                double rf = r0 + (1 - r0) * pow(1 - ct, 5.0);
                refl = rf > g.next();
            }
            V3 dir;
            if (refl) dir = reflect(ud, wn);
            else {  // refract, vec3.hpp:209-214
                double c2 = fmin(dot(-ud, wn), 1.0);
                V3 perp = ri * (ud + c2 * wn);
                V3 par = (-sqrt(fabs(1.0 - len2(perp)))) * wn;
                dir = perp + par;
            }
            V3 off = (dot(dir, rec.n) > 0) ? (0.0001 * rec.n) : (-0.0001 * rec.n);
            out.o = rec.p + off;
            out.d = dir;
            return true;
        }
        case ZR_MAT_ISOTROPIC: {  // constant_medium.hpp:14-18
            out.o = rec.p;
            out.d = ruv;
            att = tv;
            return true;
        }
        default: return false;  // diffuse_light, material.hpp:255-259
    }
}

// ---- the LEAN shading set: worlds of bare triangles and spheres whose materials are lambertian / metal / dielectric / light with solid-colour textures and
// no bump map (cfg2, cfg3: DScene::shade_lean, decided at commit).  The same arithmetic as the general routines above with everything such a world cannot
// reach left out — u, v, tangent, bitangent, image and checker textures, bump maps, wrapper chains — so that SHADE's lean build needs fewer registers:
// the kernel's speed follows its occupancy (profiles/r4_experiments_ab.txt).
struct RecL { V3 p, n; uint32_t mat; bool front; };
// x^5 for Schlick's weight (material.hpp:237-241 calls pow(1 - cosine, 5)): three multiplications, within 2 ulp of pow's result; the value only meets a
// comparison with a random number, so a different last bit changes a decision with probability ~1e-16 per dielectric hit — and libm's pow costs the
// lean build its register budget
__device__ __forceinline__ double pow5(double x) { const double x2 = x * x; return x2 * x2 * x; }
__device__ __forceinline__ void lean_rec(const DScene& sc, uint32_t kind, uint32_t idx, const Ray& r, double t, RecL& rec) {
    rec.p = at(r, t);
    V3 outward;
    bool force_front = false;
    if (kind == ZR_PRIM_SPHERE) {   // sphere.hpp:42-46
        const double* s = sc.spheres + (size_t)idx * 4;
        outward = vdiv(rec.p - mk(s[0], s[1], s[2]), s[3]);
        rec.mat = sc.sphere_mat[idx];
        if (rec.mat != 0xFFFFFFFFu && (rec.mat & 0x80000000u)) { rec.mat &= 0x7FFFFFFFu; force_front = true; }
    } else {                        // triangle.hpp:40-79
        const double* v = sc.tri_s + (size_t)idx * 20;
        const double* nn = v + 9;
        V3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
        V3 normal = cross(v1 - v0, v2 - v0);
        V3 C0 = cross(v1 - v0, rec.p - v0);
        V3 C2 = cross(v0 - v2, rec.p - v2);
        double area2 = dot(normal, normal);
        double u = dot(normal, C2) / area2;
        double w_ = dot(normal, C0) / area2;
        double w0 = 1.0 - u - w_;
        outward = unit(w0 * ld3(nn) + u * ld3(nn + 3) + w_ * ld3(nn + 6));
        rec.mat = (uint32_t)__double_as_longlong(v[18]);
        force_front = ((uint32_t)__double_as_longlong(v[19]) & 1u) != 0;
    }
    rec.front = dot(r.d, outward) < 0;   // hittable.hpp:22-25
    rec.n = rec.front ? outward : -outward;
    if (force_front) rec.front = true;
}
__device__ __forceinline__ V3 lean_color(const DScene& sc, uint32_t tex) { const zr_texture& t = sc.texs[tex]; return mk(t.color[0], t.color[1], t.color[2]); }
// emitted + scatter of the hit's material in one look at it
__device__ __forceinline__ bool lean_shade(const DScene& sc, const Ray& rin, const RecL& rec, V3& em, V3& att, Ray& out, Rng& g) {
    em = mk(0, 0, 0);
    if (rec.mat >= sc.n_mats) return false;
    const zr_material& m = sc.mats[rec.mat];
    const uint32_t kind = m.kind;
    if (kind == ZR_MAT_LIGHT) { em = lean_color(sc, m.tex); return false; }
    V3 ruv = mk(0, 0, 0);   // (one call site for both kinds that draw a unit vector first: see scatter())
    if (kind == ZR_MAT_LAMBERTIAN || kind == ZR_MAT_METAL) ruv = random_unit_vector(g);
    V3 ud = mk(0, 0, 0);
    if (kind == ZR_MAT_METAL || kind == ZR_MAT_DIELECTRIC) ud = unit(rin.d);
    if (kind == ZR_MAT_LAMBERTIAN) {  // material.hpp:74-96
        V3 dir = rec.n + ruv;
        if (fabs(dir.x) < 1e-8 && fabs(dir.y) < 1e-8 && fabs(dir.z) < 1e-8) dir = rec.n;
        out.o = rec.p + (rec.n * 0.0001);
        out.d = dir;
        att = lean_color(sc, m.tex);
        return true;
    }
    if (kind == ZR_MAT_METAL) {  // material.hpp:129-151
        V3 refl = reflect(ud, rec.n);
        V3 dir = unit(refl + (m.param * ruv));
        out.o = rec.p + (0.0001 * rec.n);
        out.d = dir;
        att = lean_color(sc, m.tex);
        return dot(dir, rec.n) > 0;
    }
    if (kind == ZR_MAT_DIELECTRIC) {  // material.hpp:192-224, 237-241
        att = mk(m.tint[0], m.tint[1], m.tint[2]);
        double ri = rec.front ? (1.0 / m.param) : m.param;
        double ct = fmin(dot(-ud, rec.n), 1.0);
        double st = sqrt(1.0 - ct * ct);
        bool refl = ri * st > 1.0;
        if (!refl) {
            double r0 = (1 - ri) / (1 + ri);
            r0 = r0 * r0;
            double rf = r0 + (1 - r0) * pow5(1 - ct);
            refl = rf > g.next();
        }
        V3 dir;
        if (refl) dir = reflect(ud, rec.n);
        else {  // refract, vec3.hpp:209-214
            double c2 = fmin(dot(-ud, rec.n), 1.0);
            V3 perp = ri * (ud + c2 * rec.n);
            V3 par = (-sqrt(fabs(1.0 - len2(perp)))) * rec.n;
            dir = perp + par;
        }
        V3 off = (dot(dir, rec.n) > 0) ? (0.0001 * rec.n) : (-0.0001 * rec.n);
        out.o = rec.p + off;
        out.d = dir;
        return true;
    }
    return false;
}

// ---- background (camera.hpp:828-925) -------------------------------------------------------------------
__device__ inline V3 background(const DScene& sc, const DEnv& env, V3 rd) {
    if (env.mode == ZR_ENV_SOLID_COLOR) return mk(env.solid[0], env.solid[1], env.solid[2]);
    V3 ud = unit(rd);
    if (env.mode == ZR_ENV_HDR_MAP) {
        if (env.hdr_tex == ZR_NO_TEXTURE) return mk(0, 0, 0);
        V3 d = ud;
        double x1 = env.cy * d.x + env.sy * d.z;
        double z1 = -env.sy * d.x + env.cy * d.z;
        d = mk(x1, d.y, z1);
        double y2 = env.cp * d.y - env.sp * d.z;
        double z2 = env.sp * d.y + env.cp * d.z;
        d = mk(d.x, y2, z2);
        double x3 = env.cr * d.x - env.sr * d.y;
        double y3 = env.sr * d.x + env.cr * d.y;
        d = mk(x3, y3, d.z);
        const double PI = 3.14159265358979323846;
        const zr_texture& t = sc.texs[env.hdr_tex];
        if (t.kind == ZR_TEX_IMAGE_F32 && t.width > 0 && t.height > 0 && t.width <= 16384 && t.height <= 16384) {
            // Only the TEXEL INDEX of (u, v) matters (nearest lookup, texture.hpp:60-66).  Get it from FP32
            // atan2f/acosf, and fall back to the FP64 functions only when the FP32 coordinate lies within
            // `guard` texels of a texel boundary (or near the poles, where acos is ill-conditioned): the FP32
            // path's absolute error is below 4e-6 rad (2 ulp functions + input rounding), i.e. < 0.011 texel at
            // 16384 texels, so inside the guarded interior both paths select the same texel.
            const float xf = (float)d.x, yf = (float)d.y, zf = (float)d.z;
            const float PIf = 3.14159265358979323846f;
            const float fi = (atan2f(zf, xf) + PIf) * (0.15915494309189535f * (float)t.width);
            const float fj = acosf(fminf(fmaxf(yf, -1.0f), 1.0f)) * (0.3183098861837907f * (float)t.height);
            const float ri = fi - floorf(fi), rj = fj - floorf(fj);
            const float guard = 0.02f;
            if (fabsf(yf) < 0.999f && ri > guard && ri < 1.0f - guard && rj > guard && rj < 1.0f - guard && fi > guard &&
                fi < (float)t.width - guard) {
                const int i = (int)fi, j = (int)fj;
                const float* px = (const float*)(sc.texels + t.texel_offset) + ((size_t)j * t.width + i) * 3;
                return mk((double)px[0], (double)px[1], (double)px[2]) * env.intensity;
            }
        }
        double phi = atan2(d.z, d.x) + PI;
        double theta = acos(clampd(d.y, -1.0, 1.0));
        return tex_value(sc, env.hdr_tex, phi / (2 * PI), theta / PI, mk(0, 0, 0)) * env.intensity;
    }
    V3 hor = mk(env.horizon[0], env.horizon[1], env.horizon[2]);
    V3 zen = mk(env.zenith[0], env.zenith[1], env.zenith[2]);
    V3 sun = mk(env.sun[0], env.sun[1], env.sun[2]);
    double a = ud.y;
    V3 sky;
    if (a > 0.0) sky = (1.0 - a) * hor + a * zen; else sky = hor * 0.1;
    V3 fin = (sky * env.sky_scale) * env.sky_exposure;
    double focus = dot(ud, sun);
    if (env.sun_on && focus > env.sun_thr) {
        double x = clampd((focus - env.sun_thr) / ((env.sun_thr + 0.0002) - env.sun_thr), 0.0, 1.0);  // smoothstep, common.hpp:87-91
        double alpha = x * x * (3 - 2 * x);
        fin = fin + mk(env.sun_add[0], env.sun_add[1], env.sun_add[2]) * alpha;
    }
    return fin;
}

// ---- camera ray (camera.hpp:784-794, 817-825) -----------------------------------------------------------
__device__ inline Ray camera_ray(const DCamera& cam, int i, int j, Rng& g) {
    double ox = g.next() - 0.5;
    double oy = g.next() - 0.5;
    V3 ps = ld3(cam.pixel00) + ((i + ox) * ld3(cam.du)) + ((j + oy) * ld3(cam.dv));
    V3 org = ld3(cam.center);
    if (cam.defocus) {
        double px, py;
        for (;;) {  // random_in_unit_disk, vec3.hpp:174-181
            px = g.range(-1, 1);
            py = g.range(-1, 1);
            if (px * px + py * py + 0.0 * 0.0 < 1) break;
        }
        org = ld3(cam.center) + (px * ld3(cam.disk_u)) + (py * ld3(cam.disk_v));
    }
    Ray r; r.o = org; r.d = ps - org;
    return r;
}

}  // namespace zr
