// zr_kernels.hip — render kernels for gfx950 (MI355X).  Replaces the reference's std::thread row-block
// sample loop (camera.hpp:404-579) and everything it calls per sample.
//
// Kernel `render_pixels` (variant 0, "pixel-group megakernel"):
//   * a group of L = min(64, 2^floor(log2 spp)) lanes owns one pixel; with spp >= 64 that is one whole
//     wave64 per pixel, so the 64 primary rays of a wave start coherent (same pixel footprint);
//   * every lane runs the complete path of its samples (s = lane, lane + L, ...) in registers;
//   * the BVH walk keeps its per-lane stack in LDS, laid out [depth][thread] so that the 64 lanes of a
//     wave hit 64 consecutive banks (no conflicts);
//   * radiance is summed per lane and reduced across the group with wave shuffles (the "warp-reduced
//     HDR accumulator"); one lane stores the pixel mean — no atomics, so the image is bit-reproducible.
#include "zr_device.h"
#include "zr_launch.h"

namespace zr {

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

// ray_color(r, depth), camera.hpp:928-986: its own L / beta start from (0, 1); the loop index restarts at 0
template <bool COUNT>
__device__ inline V3 path_radiance(const DScene& sc, const DEnv& env, Ray cur, int depth, Rng& g, uint32_t* stack, int stride, Counters& ctr,
                                   uint32_t& segments, uint32_t& hits) {
    V3 L = mk(0, 0, 0), beta = mk(1, 1, 1);
    bool missed = false;
    for (int b = 0; b < depth; b++) {
        double t; uint32_t kind, idx;
        segments++;
        bool h = closest_hit<COUNT>(sc, cur, 0.001, g, stack, stride, t, kind, idx, ctr);
        g.bounce++;
        if (!h) { missed = true; break; }
        hits++;
        Rec rec;
        object_rec(sc, kind, idx, cur, t, rec);
        L = L + beta * emitted(sc, rec);
        V3 att; Ray out;
        if (!scatter(sc, cur, rec, att, out, g)) break;
        beta = beta * att;
        cur = out;
        if (b > 10) {
            if (len(beta) < 0.0001) break;
            double p = fmax(fmax(beta.x, beta.y), beta.z);
            p = clampd(p, 0.05, 0.95);
            if (g.next() > p) break;
            beta = beta * (1 / p);
        }
    }
    if (missed) L = L + beta * background(sc, env, cur.d);
    return L;
}

// radiance of one primary sample: camera.hpp:455-461,520 + ray_color_from_hit 989-1004
template <bool COUNT>
__device__ inline V3 sample_radiance(const DScene& sc, const DCamera& cam, const DEnv& env, int i, int j, Rng& g, uint32_t* stack,
                                     int stride, Counters& ctr, uint32_t& segments, uint32_t& hits) {
    Ray r = camera_ray(cam, i, j, g);
    double t; uint32_t kind, idx;
    segments++;
    bool h = closest_hit<COUNT>(sc, r, 0.001, g, stack, stride, t, kind, idx, ctr);
    g.bounce++;
    if (!h) return background(sc, env, r.d);
    hits++;
    Rec rec;
    object_rec(sc, kind, idx, r, t, rec);
    V3 L0 = emitted(sc, rec);
    V3 att0; Ray cur;
    if (!scatter(sc, r, rec, att0, cur, g)) return L0;
    return L0 + att0 * path_radiance<COUNT>(sc, env, cur, cam.max_depth - 1, g, stack, stride, ctr, segments, hits);
}

// one primary sample with the reflection / refraction split on (camera.hpp:455-461, 490-517, 520): after the beauty path
// the first hit is scattered AGAIN, with the draws that follow on the same stream, and a second path is traced
template <bool COUNT>
__device__ inline void sample_passes(const DScene& sc, const DCamera& cam, const DEnv& env, int i, int j, Rng& g, uint32_t* stack, int stride,
                                     Counters& ctr, uint32_t& segments, uint32_t& hits, V3& beauty, V3& reflection, V3& refraction) {
    Ray r = camera_ray(cam, i, j, g);
    double t; uint32_t kind, idx;
    segments++;
    bool h = closest_hit<COUNT>(sc, r, 0.001, g, stack, stride, t, kind, idx, ctr);
    g.bounce++;
    if (!h) { beauty = beauty + background(sc, env, r.d); return; }
    hits++;
    Rec rec;
    object_rec(sc, kind, idx, r, t, rec);
    {
        V3 L0 = emitted(sc, rec);
        V3 att0; Ray cur;
        if (scatter(sc, r, rec, att0, cur, g)) L0 = L0 + att0 * path_radiance<COUNT>(sc, env, cur, cam.max_depth - 1, g, stack, stride, ctr, segments, hits);
        beauty = beauty + L0;
    }
    V3 att; Ray scattered;
    if (scatter(sc, r, rec, att, scattered, g)) {
        V3 scol = path_radiance<COUNT>(sc, env, scattered, cam.max_depth - 1, g, stack, stride, ctr, segments, hits);
        const double luma = 0.2126 * len(scol), max_luma = 2.0;   // firefly clamp, camera.hpp:499-503
        if (luma > max_luma) scol = scol * (max_luma / luma);
        const V3 reflected_dir = reflect(unit(r.d), unit(rec.n));
        const bool is_specular = dot(unit(scattered.d), reflected_dir) > 0.9;
        if (is_specular) reflection = reflection + att * scol;
        else if (dot(scattered.d, rec.n) < 0) refraction = refraction + att * scol;
    }
}

template <bool COUNT>
__global__ __launch_bounds__(ZR_BLOCK) void render_pixels(DScene sc, DCamera cam, DEnv env, uint64_t seed, WorkDesc wd, double* __restrict__ out,
                                                           unsigned long long* __restrict__ gctr) {
    __shared__ uint32_t lds_stack[ZR_STACK_DEPTH * ZR_BLOCK];
    uint32_t* stack = lds_stack + threadIdx.x;
    const int L = wd.lanes_per_pixel;
    const int groups_per_block = ZR_BLOCK / L;
    const int group = threadIdx.x / L, lane_in_group = threadIdx.x % L;
    const long long q = (long long)blockIdx.x * groups_per_block + group;  // pixel slot in the tile enumeration
    const int tpix = wd.tile_size * wd.tile_size;
    bool active = q < (long long)wd.n_tiles * tpix;
    int px = 0, py = 0;
    if (active) {
        int tile = wd.tiles[q / tpix];
        int local = (int)(q % tpix);
        px = (tile % wd.tiles_x) * wd.tile_size + local % wd.tile_size;
        py = (tile / wd.tiles_x) * wd.tile_size + local / wd.tile_size;
        active = px >= wd.x0 && px < wd.x1 && py >= wd.y0 && py < wd.y1;
    }
    V3 sum = mk(0, 0, 0);
    Counters ctr = {0, 0, 0, 0, 0};
    uint32_t segments = 0, hits = 0, nsamp = 0;
    uint64_t draws = 0;
    if (active) {
        const uint64_t pixel = (uint64_t)py * (uint64_t)cam.W + (uint64_t)px;
        for (int s = lane_in_group; s < cam.spp; s += L) {
            Rng g; g.key = zr_stream_key(seed, pixel, (uint64_t)s); g.k = 0; g.bounce = 0;
            V3 c = sample_radiance<COUNT>(sc, cam, env, px, py, g, stack, ZR_BLOCK, ctr, segments, hits);
            sum = sum + c;
            nsamp++;
            if (COUNT) draws += g.k;
        }
    }
    // group reduction (L is a power of two <= 64 and groups are wave-aligned)
    for (int m = 1; m < L; m <<= 1) {
        sum.x += shfl_xor_f64(sum.x, m);
        sum.y += shfl_xor_f64(sum.y, m);
        sum.z += shfl_xor_f64(sum.z, m);
    }
    if (active && lane_in_group == 0) {
        const double scale = 1.0 / cam.spp;  // camera.hpp:437,531
        double* o = out + ((size_t)py * cam.W + px) * 3;
        o[0] = sum.x * scale; o[1] = sum.y * scale; o[2] = sum.z * scale;
    }
    if (COUNT && active) {
        atomicAdd(&gctr[0], (unsigned long long)nsamp);
        atomicAdd(&gctr[1], (unsigned long long)segments);
        atomicAdd(&gctr[2], (unsigned long long)ctr.nodes);
        atomicAdd(&gctr[3], (unsigned long long)ctr.sph);
        atomicAdd(&gctr[4], (unsigned long long)ctr.tri);
        atomicAdd(&gctr[5], (unsigned long long)ctr.cube);
        atomicAdd(&gctr[6], (unsigned long long)ctr.med);
        atomicAdd(&gctr[7], (unsigned long long)hits);
        atomicAdd(&gctr[8], (unsigned long long)draws);
    }
}

// beauty + reflection + refraction frames (use_reflection / use_refraction on): same pixel-group layout as render_pixels
__global__ __launch_bounds__(ZR_BLOCK) void passes_pixels(DScene sc, DCamera cam, DEnv env, uint64_t seed, WorkDesc wd, double* __restrict__ out_beauty,
                                                           double* __restrict__ out_reflection, double* __restrict__ out_refraction,
                                                           unsigned long long* __restrict__ gctr) {
    __shared__ uint32_t lds_stack[ZR_STACK_DEPTH * ZR_BLOCK];
    uint32_t* stack = lds_stack + threadIdx.x;
    const int L = wd.lanes_per_pixel;
    const int groups_per_block = ZR_BLOCK / L;
    const int group = threadIdx.x / L, lane_in_group = threadIdx.x % L;
    const long long q = (long long)blockIdx.x * groups_per_block + group;
    const int tpix = wd.tile_size * wd.tile_size;
    bool active = q < (long long)wd.n_tiles * tpix;
    int px = 0, py = 0;
    if (active) {
        int tile = wd.tiles[q / tpix];
        int local = (int)(q % tpix);
        px = (tile % wd.tiles_x) * wd.tile_size + local % wd.tile_size;
        py = (tile / wd.tiles_x) * wd.tile_size + local / wd.tile_size;
        active = px >= wd.x0 && px < wd.x1 && py >= wd.y0 && py < wd.y1;
    }
    V3 sb = mk(0, 0, 0), sr = mk(0, 0, 0), sf = mk(0, 0, 0);
    Counters ctr = {0, 0, 0, 0, 0};
    uint32_t segments = 0, hits = 0, nsamp = 0;
    uint64_t draws = 0;
    if (active) {
        const uint64_t pixel = (uint64_t)py * (uint64_t)cam.W + (uint64_t)px;
        for (int s = lane_in_group; s < cam.spp; s += L) {
            Rng g; g.key = zr_stream_key(seed, pixel, (uint64_t)s); g.k = 0; g.bounce = 0;
            sample_passes<true>(sc, cam, env, px, py, g, stack, ZR_BLOCK, ctr, segments, hits, sb, sr, sf);
            nsamp++;
            draws += g.k;
        }
    }
    for (int m = 1; m < L; m <<= 1) {
        sb.x += shfl_xor_f64(sb.x, m); sb.y += shfl_xor_f64(sb.y, m); sb.z += shfl_xor_f64(sb.z, m);
        sr.x += shfl_xor_f64(sr.x, m); sr.y += shfl_xor_f64(sr.y, m); sr.z += shfl_xor_f64(sr.z, m);
        sf.x += shfl_xor_f64(sf.x, m); sf.y += shfl_xor_f64(sf.y, m); sf.z += shfl_xor_f64(sf.z, m);
    }
    if (active && lane_in_group == 0) {
        const double scale = 1.0 / cam.spp;  // light_scale, camera.hpp:436-437, 531-533
        const size_t o = ((size_t)py * cam.W + px) * 3;
        if (out_beauty) { out_beauty[o] = sb.x * scale; out_beauty[o + 1] = sb.y * scale; out_beauty[o + 2] = sb.z * scale; }
        if (out_reflection) { out_reflection[o] = sr.x * scale; out_reflection[o + 1] = sr.y * scale; out_reflection[o + 2] = sr.z * scale; }
        if (out_refraction) { out_refraction[o] = sf.x * scale; out_refraction[o + 1] = sf.y * scale; out_refraction[o + 2] = sf.z * scale; }
    }
    if (active) {
        atomicAdd(&gctr[0], (unsigned long long)nsamp);
        atomicAdd(&gctr[1], (unsigned long long)segments);
        atomicAdd(&gctr[7], (unsigned long long)hits);
        atomicAdd(&gctr[8], (unsigned long long)draws);
    }
}

// first-hit AOVs (camera.hpp:433, 464-488, 521-541): same pixel-group layout as render_pixels, primary rays only
__global__ __launch_bounds__(ZR_BLOCK) void aov_pixels(DScene sc, DCamera cam, uint64_t seed, WorkDesc wd, int aux, double zmax, double3 cu, double3 cv,
                                                        double3 cw, double* __restrict__ out_albedo, double* __restrict__ out_normal,
                                                        double* __restrict__ out_zdepth) {
    __shared__ uint32_t lds_stack[ZR_STACK_DEPTH * ZR_BLOCK];
    uint32_t* stack = lds_stack + threadIdx.x;
    const int L = wd.lanes_per_pixel;
    const int groups_per_block = ZR_BLOCK / L;
    const int group = threadIdx.x / L, lane_in_group = threadIdx.x % L;
    const long long q = (long long)blockIdx.x * groups_per_block + group;
    const int tpix = wd.tile_size * wd.tile_size;
    bool active = q < (long long)wd.n_tiles * tpix;
    int px = 0, py = 0;
    if (active) {
        int tile = wd.tiles[q / tpix];
        int local = (int)(q % tpix);
        px = (tile % wd.tiles_x) * wd.tile_size + local % wd.tile_size;
        py = (tile / wd.tiles_x) * wd.tile_size + local / wd.tile_size;
        active = px >= wd.x0 && px < wd.x1 && py >= wd.y0 && py < wd.y1;
    }
    V3 a = mk(0, 0, 0), n = mk(0, 0, 0), z = mk(0, 0, 0);
    Counters ctr = {0, 0, 0, 0, 0};
    if (active) {
        const uint64_t pixel = (uint64_t)py * (uint64_t)cam.W + (uint64_t)px;
        for (int s = lane_in_group; s < aux; s += L) {
            Rng g; g.key = zr_stream_key(seed, pixel, (uint64_t)s); g.k = 0; g.bounce = 0;
            Ray r = camera_ray(cam, px, py, g);
            double t; uint32_t kind, idx;
            if (closest_hit<false>(sc, r, 0.001, g, stack, ZR_BLOCK, t, kind, idx, ctr)) {
                Rec rec;
                object_rec(sc, kind, idx, r, t, rec);
                a = a + get_albedo(sc, rec);
                V3 un = unit(rec.n);
                n = n + mk((dot(un, mk(cu.x, cu.y, cu.z)) + 1.0) * 0.5, (dot(un, mk(cv.x, cv.y, cv.z)) + 1.0) * 0.5, (dot(un, mk(cw.x, cw.y, cw.z)) + 1.0) * 0.5);
                double zd = 1.0 - clampd(rec.t / zmax, 0.0, 1.0);
                z = z + mk(zd, zd, zd);
            } else {
                n = n + mk(0.5, 0.5, 1.0);
            }
        }
    }
    for (int m = 1; m < L; m <<= 1) {
        a.x += shfl_xor_f64(a.x, m); a.y += shfl_xor_f64(a.y, m); a.z += shfl_xor_f64(a.z, m);
        n.x += shfl_xor_f64(n.x, m); n.y += shfl_xor_f64(n.y, m); n.z += shfl_xor_f64(n.z, m);
        z.x += shfl_xor_f64(z.x, m); z.y += shfl_xor_f64(z.y, m); z.z += shfl_xor_f64(z.z, m);
    }
    if (active && lane_in_group == 0) {
        const double scale = 1.0 / aux;  // camera.hpp:535-536
        const size_t o = ((size_t)py * cam.W + px) * 3;
        if (out_albedo) { out_albedo[o] = a.x * scale; out_albedo[o + 1] = a.y * scale; out_albedo[o + 2] = a.z * scale; }
        if (out_normal) { out_normal[o] = n.x * scale; out_normal[o + 1] = n.y * scale; out_normal[o + 2] = n.z * scale; }
        if (out_zdepth) { out_zdepth[o] = z.x * scale; out_zdepth[o + 1] = z.y * scale; out_zdepth[o + 2] = z.z * scale; }
    }
}

// known-answer kernel: world.hit(r, interval(tmin, tmax), rec) for a batch of rays, one ray per thread
__global__ __launch_bounds__(ZR_BLOCK) void trace_rays(DScene sc, const double* __restrict__ rays, size_t n, double tmin, double tmax,
                                                        uint64_t seed, uint64_t pixel, uint32_t bounce, zr_hit* __restrict__ out) {
    __shared__ uint32_t lds_stack[ZR_STACK_DEPTH * ZR_BLOCK];
    size_t k = (size_t)blockIdx.x * ZR_BLOCK + threadIdx.x;
    if (k >= n) return;
    Ray r; r.o = ld3(rays + k * 6); r.d = ld3(rays + k * 6 + 3);
    Rng g; g.key = zr_stream_key(seed, pixel, k); g.k = 0; g.bounce = bounce;
    Counters ctr = {0, 0, 0, 0, 0};
    double t; uint32_t kind, idx;
    bool h = closest_hit<false>(sc, r, tmin, g, lds_stack + threadIdx.x, ZR_BLOCK, t, kind, idx, ctr, tmax);
    zr_hit o;
    if (h) {
        Rec rec;
        object_rec(sc, kind, idx, r, t, rec, true);
        o.p[0] = rec.p.x; o.p[1] = rec.p.y; o.p[2] = rec.p.z;
        o.normal[0] = rec.n.x; o.normal[1] = rec.n.y; o.normal[2] = rec.n.z;
        o.tangent[0] = rec.tan.x; o.tangent[1] = rec.tan.y; o.tangent[2] = rec.tan.z;
        o.bitangent[0] = rec.bit.x; o.bitangent[1] = rec.bit.y; o.bitangent[2] = rec.bit.z;
        o.t = rec.t; o.u = rec.u; o.v = rec.v; o.mat = rec.mat; o.front_face = rec.front ? 1u : 0u;
    } else {
        for (int c = 0; c < 3; c++) { o.p[c] = 0; o.normal[c] = 0; o.tangent[c] = 0; o.bitangent[c] = 0; }
        o.t = 0; o.u = 0; o.v = 0; o.mat = 0xFFFFFFFFu; o.front_face = 0;
    }
    out[k] = o;
}

// ---- per-function known-answer kernels (zr_kat_*): one thread = one call of the reference's virtual ---------------------
__global__ __launch_bounds__(ZR_BLOCK) void kat_scatter(DScene sc, const double* __restrict__ rays, const zr_hit* __restrict__ recs,
                                                         const uint64_t* __restrict__ keys, const uint64_t* __restrict__ first_draw, size_t n,
                                                         zr_scatter_out* __restrict__ out) {
    size_t k = (size_t)blockIdx.x * ZR_BLOCK + threadIdx.x;
    if (k >= n) return;
    Ray r; r.o = ld3(rays + k * 6); r.d = ld3(rays + k * 6 + 3);
    const zr_hit h = recs[k];
    Rec rec;
    rec.p = mk(h.p[0], h.p[1], h.p[2]); rec.n = mk(h.normal[0], h.normal[1], h.normal[2]);
    rec.tan = mk(h.tangent[0], h.tangent[1], h.tangent[2]); rec.bit = mk(h.bitangent[0], h.bitangent[1], h.bitangent[2]);
    rec.t = h.t; rec.u = h.u; rec.v = h.v; rec.mat = h.mat; rec.front = h.front_face != 0;
    Rng g; g.key = keys[k]; g.k = first_draw ? first_draw[k] : 0; g.bounce = 0;
    const uint64_t k0 = g.k;
    V3 att = mk(0, 0, 0); Ray nr; nr.o = mk(0, 0, 0); nr.d = mk(0, 0, 0);
    const V3 em = emitted(sc, rec);                         // material::emitted, material.hpp:12-14,261-263
    const bool ok = scatter(sc, r, rec, att, nr, g);        // material::scatter
    zr_scatter_out o;
    o.attenuation[0] = att.x; o.attenuation[1] = att.y; o.attenuation[2] = att.z;
    o.origin[0] = nr.o.x; o.origin[1] = nr.o.y; o.origin[2] = nr.o.z;
    o.direction[0] = nr.d.x; o.direction[1] = nr.d.y; o.direction[2] = nr.d.z;
    o.emitted[0] = em.x; o.emitted[1] = em.y; o.emitted[2] = em.z;
    o.scattered = ok ? 1u : 0u; o.draws = (uint32_t)(g.k - k0);
    if (!ok) for (int c = 0; c < 3; c++) { o.attenuation[c] = 0; o.origin[c] = 0; o.direction[c] = 0; }
    out[k] = o;
}

__global__ __launch_bounds__(ZR_BLOCK) void kat_texture(DScene sc, uint32_t tex, const double* __restrict__ uvp, size_t n, double* __restrict__ out) {
    size_t k = (size_t)blockIdx.x * ZR_BLOCK + threadIdx.x;
    if (k >= n) return;
    const V3 c = tex_value(sc, tex, uvp[k * 5], uvp[k * 5 + 1], mk(uvp[k * 5 + 2], uvp[k * 5 + 3], uvp[k * 5 + 4]));   // texture::value
    out[k * 3] = c.x; out[k * 3 + 1] = c.y; out[k * 3 + 2] = c.z;
}

__global__ __launch_bounds__(ZR_BLOCK) void kat_background(DScene sc, DEnv env, const double* __restrict__ dirs, size_t n, double* __restrict__ out) {
    size_t k = (size_t)blockIdx.x * ZR_BLOCK + threadIdx.x;
    if (k >= n) return;
    const V3 c = background(sc, env, ld3(dirs + k * 3));     // camera::get_background_color
    out[k * 3] = c.x; out[k * 3 + 1] = c.y; out[k * 3 + 2] = c.z;
}

__global__ __launch_bounds__(ZR_BLOCK) void kat_camera_rays(DCamera cam, uint64_t seed, const int32_t* __restrict__ req, size_t n, double* __restrict__ out) {
    size_t k = (size_t)blockIdx.x * ZR_BLOCK + threadIdx.x;
    if (k >= n) return;
    const int px = req[k * 3], py = req[k * 3 + 1], smp = req[k * 3 + 2];
    Rng g; g.key = zr_stream_key(seed, (uint64_t)py * (uint64_t)cam.W + (uint64_t)px, (uint64_t)smp); g.k = 0; g.bounce = 0;
    const Ray r = camera_ray(cam, px, py, g);               // camera::get_ray
    out[k * 7] = r.o.x; out[k * 7 + 1] = r.o.y; out[k * 7 + 2] = r.o.z; out[k * 7 + 3] = r.d.x; out[k * 7 + 4] = r.d.y; out[k * 7 + 5] = r.d.z;
    out[k * 7 + 6] = (double)g.k;
}

// ---- known-answer kernel for whole paths: one thread walks one primary sample and records every segment ------------
// record (ZR_PATH_REC doubles per segment): ray o, d | hit flag, t, material | scattered flag, attenuation rgb |
// emission rgb | main-stream draws consumed after this segment
__global__ __launch_bounds__(ZR_BLOCK) void path_records(DScene sc, DCamera cam, uint64_t seed, const int32_t* __restrict__ req, int n_req, int max_seg,
                                                          double* __restrict__ out) {
    __shared__ uint32_t lds_stack[ZR_STACK_DEPTH * ZR_BLOCK];
    const int q = blockIdx.x * ZR_BLOCK + threadIdx.x;
    if (q >= n_req) return;
    uint32_t* stack = lds_stack + threadIdx.x;
    const int px = req[q * 3], py = req[q * 3 + 1], smp = req[q * 3 + 2];
    double* rec_out = out + (size_t)q * max_seg * ZR_PATH_REC;
    for (int k = 0; k < max_seg * ZR_PATH_REC; k++) rec_out[k] = 0.0;
    Rng g; g.key = zr_stream_key(seed, (uint64_t)py * (uint64_t)cam.W + (uint64_t)px, (uint64_t)smp); g.k = 0; g.bounce = 0;
    Counters ctr = {0, 0, 0, 0, 0};
    Ray cur = camera_ray(cam, px, py, g);
    V3 beta = mk(1, 1, 1);
    int inner = -1;  // -1: the peeled first bounce (ray_color_from_hit), then ray_color's loop index
    for (int seg = 0; seg < max_seg && seg < cam.max_depth; seg++) {
        double* o = rec_out + (size_t)seg * ZR_PATH_REC;
        o[0] = cur.o.x; o[1] = cur.o.y; o[2] = cur.o.z; o[3] = cur.d.x; o[4] = cur.d.y; o[5] = cur.d.z;
        double t; uint32_t kind, idx;
        bool h = closest_hit<false>(sc, cur, 0.001, g, stack, ZR_BLOCK, t, kind, idx, ctr);
        g.bounce++;
        if (!h) { o[6] = 0; o[16] = (double)g.k; break; }
        Rec rec;
        object_rec(sc, kind, idx, cur, t, rec);
        V3 em = emitted(sc, rec);
        V3 att; Ray nxt;
        bool ok = scatter(sc, cur, rec, att, nxt, g);
        o[6] = 1; o[7] = t; o[8] = (double)rec.mat; o[9] = ok ? 1 : 0;
        o[10] = ok ? att.x : 0; o[11] = ok ? att.y : 0; o[12] = ok ? att.z : 0; o[13] = em.x; o[14] = em.y; o[15] = em.z;
        if (!ok) { o[16] = (double)g.k; break; }
        beta = beta * att;
        cur = nxt;
        if (inner > 10) {   // camera.hpp:972-980, loop index of ray_color
            if (len(beta) < 0.0001) { o[16] = (double)g.k; break; }
            double p = fmax(fmax(beta.x, beta.y), beta.z);
            p = clampd(p, 0.05, 0.95);
            if (g.next() > p) { o[16] = (double)g.k; break; }
            beta = beta * (1 / p);
        }
        if (inner < 0) beta = mk(1, 1, 1);   // ray_color starts its own beta after the peeled bounce
        inner++;
        o[16] = (double)g.k;
    }
}

// ---- launch wrappers (called from zr_render.cpp) -------------------------------------------------------
hipError_t launch_render(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, const WorkDesc& wd, double* out,
                         unsigned long long* gctr, bool count, hipStream_t stream) {
    const int groups_per_block = ZR_BLOCK / wd.lanes_per_pixel;
    const long long pixels = (long long)wd.n_tiles * wd.tile_size * wd.tile_size;
    const long long blocks = (pixels + groups_per_block - 1) / groups_per_block;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block(ZR_BLOCK);
    if (count) hipLaunchKernelGGL(render_pixels<true>, grid, block, 0, stream, sc, cam, env, seed, wd, out, gctr);
    else hipLaunchKernelGGL(render_pixels<false>, grid, block, 0, stream, sc, cam, env, seed, wd, out, gctr);
    return hipGetLastError();
}

hipError_t launch_aov(const DScene& sc, const DCamera& cam, uint64_t seed, const WorkDesc& wd, int aux, double zmax, const double* uvw9,
                      double* out_albedo, double* out_normal, double* out_zdepth, hipStream_t stream) {
    const int groups_per_block = ZR_BLOCK / wd.lanes_per_pixel;
    const long long pixels = (long long)wd.n_tiles * wd.tile_size * wd.tile_size;
    const long long blocks = (pixels + groups_per_block - 1) / groups_per_block;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    double3 cu = make_double3(uvw9[0], uvw9[1], uvw9[2]), cv = make_double3(uvw9[3], uvw9[4], uvw9[5]), cw = make_double3(uvw9[6], uvw9[7], uvw9[8]);
    hipLaunchKernelGGL(aov_pixels, dim3((unsigned)blocks), dim3(ZR_BLOCK), 0, stream, sc, cam, seed, wd, aux, zmax, cu, cv, cw, out_albedo, out_normal, out_zdepth);
    return hipGetLastError();
}

hipError_t launch_passes(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, const WorkDesc& wd, double* out_beauty,
                         double* out_reflection, double* out_refraction, unsigned long long* gctr, hipStream_t stream) {
    const int groups_per_block = ZR_BLOCK / wd.lanes_per_pixel;
    const long long pixels = (long long)wd.n_tiles * wd.tile_size * wd.tile_size;
    const long long blocks = (pixels + groups_per_block - 1) / groups_per_block;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(passes_pixels, dim3((unsigned)blocks), dim3(ZR_BLOCK), 0, stream, sc, cam, env, seed, wd, out_beauty, out_reflection, out_refraction, gctr);
    return hipGetLastError();
}

hipError_t launch_path_records(const DScene& sc, const DCamera& cam, uint64_t seed, const int32_t* req, int n_req, int max_seg, double* out,
                               hipStream_t stream) {
    if (n_req <= 0) return hipSuccess;
    hipLaunchKernelGGL(path_records, dim3((unsigned)((n_req + ZR_BLOCK - 1) / ZR_BLOCK)), dim3(ZR_BLOCK), 0, stream, sc, cam, seed, req, n_req, max_seg, out);
    return hipGetLastError();
}

hipError_t launch_kat_scatter(const DScene& sc, const double* rays, const zr_hit* recs, const uint64_t* keys, const uint64_t* first_draw, size_t n,
                              zr_scatter_out* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kat_scatter, dim3((unsigned)((n + ZR_BLOCK - 1) / ZR_BLOCK)), dim3(ZR_BLOCK), 0, stream, sc, rays, recs, keys, first_draw, n, out);
    return hipGetLastError();
}
hipError_t launch_kat_texture(const DScene& sc, uint32_t tex, const double* uvp, size_t n, double* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kat_texture, dim3((unsigned)((n + ZR_BLOCK - 1) / ZR_BLOCK)), dim3(ZR_BLOCK), 0, stream, sc, tex, uvp, n, out);
    return hipGetLastError();
}
hipError_t launch_kat_background(const DScene& sc, const DEnv& env, const double* dirs, size_t n, double* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kat_background, dim3((unsigned)((n + ZR_BLOCK - 1) / ZR_BLOCK)), dim3(ZR_BLOCK), 0, stream, sc, env, dirs, n, out);
    return hipGetLastError();
}
hipError_t launch_kat_camera_rays(const DCamera& cam, uint64_t seed, const int32_t* req, size_t n, double* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kat_camera_rays, dim3((unsigned)((n + ZR_BLOCK - 1) / ZR_BLOCK)), dim3(ZR_BLOCK), 0, stream, cam, seed, req, n, out);
    return hipGetLastError();
}

hipError_t launch_trace(const DScene& sc, const double* rays, size_t n, double tmin, double tmax, uint64_t seed, uint64_t pixel,
                        uint32_t bounce, zr_hit* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 grid((unsigned)((n + ZR_BLOCK - 1) / ZR_BLOCK)), block(ZR_BLOCK);
    hipLaunchKernelGGL(trace_rays, grid, block, 0, stream, sc, rays, n, tmin, tmax, seed, pixel, bounce, out);
    return hipGetLastError();
}

}  // namespace zr
