// ref_harness.cpp — TEST INFRASTRUCTURE (oracle).  Never shipped, never on the product path.
//
// Drives the GENUINE reference arithmetic from /root/reference (headers are #included by path at build
// time, in this container only; nothing of them is copied into the repo):
//   model (OBJ ingest through the vendored tiny_obj_loader)   model.hpp:12-107
//   world.hit          bvh.hpp:46-54,112-118  aabb.hpp:44-66  sphere.hpp:18-64  triangle.hpp:17-82
//                      cube.hpp:44-142  constant_medium.hpp:39-77  translate/rotate_*/scale/material_instance
//   scatter / emitted  material.hpp:74-96,129-151,192-224,261-263  constant_medium.hpp:14-18
//   texture::value     texture.hpp:50-78,96-98,118-126
//   random_double() and the rejection samplers  common.hpp:29-44  vec3.hpp:174-191
//
// camera.hpp cannot be #included in this image — it includes <OpenImageDenoise/oidn.hpp> (camera.hpp:10), a library
// that is absent, and no stand-in header is written for it.  Its hot-path MEMBER FUNCTIONS are nevertheless the genuine
// ones: oracle/Makefile cuts their text out of camera.hpp at build time and class ref_camera below includes it (see
// there).  Only the body of the per-sample loop of render_rows (camera.hpp:454-531, a lambda that cannot be cut out)
// is restated, in ref_camera::sample / aov_sample / passes_sample.
//
// Determinism: ref_prelude.hpp injects the counter engine of include/zr_rng.h in place of the
// reference's racy global mt19937.  Build with ROCm clang++ (left-to-right argument evaluation, the
// same front end as hipcc): see oracle/Makefile.
#include "ref_prelude.hpp"

#include <sys/stat.h>
#include <unistd.h>
#include <chrono>

#include "../include/zr_capi.h"

// ---- genuine reference headers (order as main.cpp:7-13 needs it) -----------------------------
#include "common.hpp"
#include "hittable.hpp"
#include "texture.hpp"
#include "material.hpp"
#include "hittable_list.hpp"
#include "sphere.hpp"
#include "triangle.hpp"
#include "cube.hpp"
#include "constant_medium.hpp"
#include "translate.hpp"
#include "rotate_x.hpp"
#include "rotate_y.hpp"
#include "rotate_z.hpp"
#include "scale.hpp"
#include "material_instance.hpp"
#include "bvh.hpp"
#include "model.hpp"
#include "environment.hpp"
#include "bloom.hpp"             // genuine post stack
#include "color_processing.hpp"

// ---- hooks for scenes/zr_scenes.inc ------------------------------------------------------------
static void zr_hook_seed_scene(uint64_t seed, uint64_t stream) { zr_oracle_seed(seed, ZR_SCENE_PIXEL, stream); }

// tags a constant_medium with its id so that the draw inside constant_medium::hit is keyed off-stream
class keyed_medium : public hittable {
public:
    keyed_medium(shared_ptr<hittable> m, uint32_t id) : inner(m), id(id) {}
    bool hit(const ray& r, interval ray_t, hit_record& rec, int depth = 0, bool debug_wire = false) const override {
        zr_oracle_tls_t& s = zr_oracle_tls;
        s.in_medium = true; s.medium_id = id;
        bool h = inner->hit(r, ray_t, rec, depth, debug_wire);
        s.in_medium = false;
        return h;
    }
    aabb bounding_box() const override { return inner->bounding_box(); }
private:
    shared_ptr<hittable> inner; uint32_t id;
};
static uint32_t g_next_medium_id = 0;
static shared_ptr<hittable> zr_hook_medium(shared_ptr<hittable> m) { return make_shared<keyed_medium>(m, g_next_medium_id++); }

#include "../scenes/zr_scenes.inc"
#include "../scenes/zr_scenes_mix.inc"
#include "scene_management.hpp"             // the reference's own demo scene: load_materials / sceneAssetsLoader / build_geometry
#include "../scenes/zr_scene_refdemo.inc"

// counts closest-hit queries ("segments") and publishes the bounce index for the medium key
struct seg_tls_t { uint64_t segments = 0; };
static thread_local seg_tls_t seg_tls;
class probe_world : public hittable {
public:
    explicit probe_world(const hittable& w) : w(w) {}
    bool hit(const ray& r, interval ray_t, hit_record& rec, int depth = 0, bool debug_wire = false) const override {
        zr_oracle_tls.bounce = (uint32_t)seg_in_sample();
        seg_tls.segments++;
        sample_segments()++;
        const bool h = w.hit(r, ray_t, rec, depth, debug_wire);
        // diagnostic (ZR_REF_TRACE=1 with a one-pixel tile): every closest-hit query of the genuine code, as it happens
        static const bool trace = std::getenv("ZR_REF_TRACE") != nullptr;
        if (trace) std::fprintf(stderr, "[ref] seg %llu draws %llu o %.17g %.17g %.17g d %.17g %.17g %.17g -> hit %d t %.17g p %.9g %.9g %.9g n %.6g %.6g %.6g\n",
                                (unsigned long long)sample_segments(), (unsigned long long)zr_oracle_tls.k, r.origin().x(), r.origin().y(), r.origin().z(), r.direction().x(),
                                r.direction().y(), r.direction().z(), (int)h, h ? rec.t : 0.0, h ? rec.p.x() : 0.0, h ? rec.p.y() : 0.0, h ? rec.p.z() : 0.0,
                                h ? rec.normal.x() : 0.0, h ? rec.normal.y() : 0.0, h ? rec.normal.z() : 0.0);
        return h;
    }
    aabb bounding_box() const override { return w.bounding_box(); }
    static uint64_t& sample_segments() { static thread_local uint64_t n = 0; return n; }
    static uint64_t seg_in_sample() { return sample_segments(); }
private:
    const hittable& w;
};

// ---- the reference's camera member functions, compiled from camera.hpp itself ------------------------------
// camera.hpp cannot be #included (OpenImageDenoise, line 10).  oracle/Makefile therefore cuts the TEXT of the member
// functions on the hot path out of /root/reference/camera.hpp at build time — initialize() (camera.hpp:358-399, i.e.
// without its last statement, the refresh_hdr_list() directory scan of line 401) and get_ray ... ray_color_from_hit
// (camera.hpp:784-1004) — into a scratch file outside the repo (zr_ref_camera_members.inc in a mktemp directory that
// the recipe deletes after the compile: reference text is neither committed nor shipped to the GPU box) and this
// scaffold class, which declares nothing but the data members those functions read (names and defaults as
// camera.hpp:26-44,341-349), includes it.  So initialize / get_ray / sample_square / defocus_disk_sample /
// get_background_color / ray_color / ray_color_from_hit below are the GENUINE functions, not restatements.
class ref_camera {
public:
    double aspect_ratio = 1.0;
    int image_width = 400, image_height = 225, samples_per_pixel = 30, max_depth = 10;
    double vfov = 30;
    point3 lookfrom = point3(10, 1.5, 0), lookat = point3(0, 0, 0);
    vec3 vup = vec3(0, 1, 0);
    double defocus_angle = 0.5, focus_dist = 10;
    double pixel_samples_scale = 0.0;
    point3 center, pixel00_loc;
    vec3 pixel_delta_u, pixel_delta_v, u, v, w, defocus_disk_u, defocus_disk_v;

#include "zr_ref_camera_members.inc"

    // ---- harness side: everything below is the oracle's own code -------------------------------------------
    zr_camera c{};
    void setup() {   // C-ABI camera description -> the reference's public fields, then the genuine initialize()
        image_width = c.image_width; image_height = c.image_height; samples_per_pixel = c.samples_per_pixel; max_depth = c.max_depth;
        vfov = c.vfov;
        lookfrom = point3(c.lookfrom[0], c.lookfrom[1], c.lookfrom[2]);
        lookat = point3(c.lookat[0], c.lookat[1], c.lookat[2]);
        vup = vec3(c.vup[0], c.vup[1], c.vup[2]);
        defocus_angle = c.defocus_angle; focus_dist = c.focus_dist;
        initialize();
    }

    // The three functions below restate the BODY of the per-sample loop of render_rows (camera.hpp:454-531), which is a
    // lambda inside execute_render_threads and cannot be cut out: every call they make is to genuine code.
    // first-hit AOVs of one primary sample: camera.hpp:464-488 (hit) and 521-525 (miss)
    void aov_sample(int i, int j, const hittable& world, double zmax, color& albedo, color& normal, color& zdepth) const {
        ray r = get_ray(i, j);
        hit_record rec;
        if (world.hit(r, interval(0.001, infinity), rec)) {
            albedo += rec.mat->get_albedo(rec);
            vec3 n = unit_vector(rec.normal);
            double nx = dot(n, u), ny = dot(n, v), nz = dot(n, w);
            normal += color((nx + 1.0) * 0.5, (ny + 1.0) * 0.5, (nz + 1.0) * 0.5);
            double zd = 1.0 - std::clamp(rec.t / zmax, 0.0, 1.0);
            zdepth += color(zd, zd, zd);
        } else {
            normal += color(0.5, 0.5, 1.0);
        }
    }

    // one primary sample with the reflection / refraction split enabled: camera.hpp:455-461 (beauty), 490-517 (the
    // first hit is scattered AGAIN, with fresh draws, and a second path is traced), 520 (miss)
    void passes_sample(int i, int j, const hittable& world, const EnvironmentSettings& env, color& beauty, color& reflection,
                       color& refraction) const {
        ray r = get_ray(i, j);
        hit_record rec;
        if (world.hit(r, interval(0.001, infinity), rec)) {
            beauty += ray_color_from_hit(r, rec, world, max_depth, env);
            ray scattered;
            color attenuation;
            if (rec.mat->scatter(r, rec, attenuation, scattered)) {
                color scattered_color = ray_color(scattered, world, max_depth - 1, env);
                double luma = 0.2126 * scattered_color.length();
                double max_luma = 2.0;
                if (luma > max_luma) scattered_color *= (max_luma / luma);
                vec3 reflected_dir = reflect(unit_vector(r.direction()), unit_vector(rec.normal));
                bool is_specular = dot(unit_vector(scattered.direction()), reflected_dir) > 0.9;
                if (is_specular) reflection += attenuation * scattered_color;
                else if (dot(scattered.direction(), rec.normal) < 0) refraction += attenuation * scattered_color;
            }
        } else {
            beauty += get_background_color(r, env);
        }
    }

    // one primary sample of pixel (i, j): the body of the sample loop, camera.hpp:455-461,520
    color sample(int i, int j, const hittable& world, const EnvironmentSettings& env) const {
        ray r = get_ray(i, j);
        hit_record rec;
        if (world.hit(r, interval(0.001, infinity), rec)) return ray_color_from_hit(r, rec, world, max_depth, env);
        return get_background_color(r, env);
    }
};

// ---- tiny .npy writer --------------------------------------------------------------------------
static void write_npy(const std::string& path, const char* descr, const std::vector<size_t>& shape, const void* data,
                      size_t bytes) {
    std::string sh = "(";
    for (size_t k = 0; k < shape.size(); k++) { sh += std::to_string(shape[k]); sh += (shape.size() == 1 || k + 1 < shape.size()) ? "," : ""; }
    sh += ")";
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + sh + ", }";
    size_t total = 10 + hdr.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    hdr += std::string(pad, ' ');
    hdr += "\n";
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    unsigned char magic[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hdr.size() & 255), (unsigned char)(hdr.size() >> 8)};
    std::fwrite(magic, 1, 10, f);
    std::fwrite(hdr.data(), 1, hdr.size(), f);
    std::fwrite(data, 1, bytes, f);
    std::fclose(f);
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct built_scene {
    zr_demo_scene s;
    shared_ptr<hittable> bvh;
    double build_s = 0;
};

static bool build(built_scene& b, const std::string& name, int a0, int a1, int a2, int a3) {
    g_next_medium_id = 0;
    bool ok = zr_build_scene(name, b.s, a0, a1, a2, a3) || zr_build_scene_mix(name, b.s, a0);
    if (!ok && name == "refdemo") { zr_build_refdemo(b.s); ok = true; }   // needs the reference's assets/ tree in the working directory
    if (!ok) return false;
    // bvh_node's constructor draws random_int(0,2) per node (bvh.hpp:17): give it its own scene stream
    zr_oracle_seed(b.s.seed, ZR_SCENE_PIXEL, 1);
    double t0 = now_s();
    b.bvh = make_shared<bvh_node>(b.s.world);
    b.build_s = now_s() - t0;
    return true;
}

static void cleanup(built_scene& b) { for (auto& f : b.s.temp_files) ::unlink(f.c_str()); }

// render [x0,x0+w) x [y0,y0+h) with `threads` row-interleaved host threads.
// out: mean radiance (h*w*3 doubles); optional per-sample radiance and per-sample draw/segment counts.
struct tile_result { uint64_t segments = 0, draws = 0, medium_draws = 0; double seconds = 0; };
static tile_result render_tile(const built_scene& b, ref_camera cam, int x0, int y0, int w, int h, int spp, int threads,
                               int xstep, int ystep, std::vector<double>& mean, std::vector<double>* per_sample,
                               std::vector<uint32_t>* per_sample_counts) {
    cam.c.samples_per_pixel = spp;
    cam.setup();
    const int W = cam.c.image_width;
    mean.assign((size_t)w * h * 3, 0.0);
    if (per_sample) per_sample->assign((size_t)w * h * spp * 3, 0.0);
    if (per_sample_counts) per_sample_counts->assign((size_t)w * h * spp * 2, 0);
    probe_world world(*b.bvh);
    std::atomic<uint64_t> segs{0}, draws{0}, mdraws{0};
    std::atomic<int> next_row{0};
    double t0 = now_s();
    auto worker = [&]() {
        seg_tls.segments = 0; zr_oracle_tls.draws = 0; zr_oracle_tls.medium_draws = 0;
        for (;;) {
            int jj = next_row.fetch_add(1);
            if (jj >= h) break;
            int j = y0 + jj * ystep;
            for (int ii = 0; ii < w; ii++) {
                int i = x0 + ii * xstep;
                color acc(0, 0, 0);
                for (int s = 0; s < spp; s++) {
                    zr_oracle_seed(b.s.seed, (uint64_t)j * W + i, (uint64_t)s);
                    probe_world::sample_segments() = 0;
                    uint64_t d0 = zr_oracle_tls.draws;
                    color c = cam.sample(i, j, world, b.s.env);
                    acc += c;
                    if (per_sample) {
                        double* o = &(*per_sample)[(((size_t)jj * w + ii) * spp + s) * 3];
                        o[0] = c.x(); o[1] = c.y(); o[2] = c.z();
                    }
                    if (per_sample_counts) {
                        uint32_t* o = &(*per_sample_counts)[(((size_t)jj * w + ii) * spp + s) * 2];
                        o[0] = (uint32_t)probe_world::sample_segments();
                        o[1] = (uint32_t)(zr_oracle_tls.draws - d0);
                    }
                }
                color m = acc * (1.0 / spp);  // camera.hpp:436-437,531 (light_scale)
                double* o = &mean[((size_t)jj * w + ii) * 3];
                o[0] = m.x(); o[1] = m.y(); o[2] = m.z();
            }
        }
        segs += seg_tls.segments; draws += zr_oracle_tls.draws; mdraws += zr_oracle_tls.medium_draws;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    tile_result r;
    r.seconds = now_s() - t0; r.segments = segs; r.draws = draws; r.medium_draws = mdraws;
    return r;
}

static int usage() {
    std::fprintf(stderr,
                 "usage:\n"
                 "  zenith_ref tile   <scene> <x0> <y0> <w> <h> <spp|0> <threads> <out_prefix> [per_sample=0] [a0 a1 a2 a3]\n"
                 "  zenith_ref time   <scene> <xstep> <ystep> <spp|0> <threads> [a0 a1 a2 a3]\n"
                 "  zenith_ref trace  <scene> <nrays> <seed> <out_prefix> <clamp_lo> <clamp_hi> [a0 a1 a2 a3]\n"
                 "  zenith_ref aov    <scene> <x0> <y0> <w> <h> <zmax> <out_prefix> - [a0 a1 a2 a3]\n"
                 "  zenith_ref passes <scene> <x0> <y0> <w> <h> <spp|0> <out_prefix> [a0 a1 a2 a3]\n"
                 "  zenith_ref post   <scene> <preset 0-7> <out_prefix> [spp]\n"
                 "  zenith_ref kat    <scene> hits|tex|bg|cam <in.bin> <n> <out_prefix>\n"
                 "  zenith_ref texels <w> <h> <out.npy>\n");
    return 2;
}

int main(int argc, char** argv) {
    if (argc < 2) return usage();
    std::string cmd = argv[1];
    auto iarg = [&](int k, int dflt) { return argc > k ? std::atoi(argv[k]) : dflt; };

    if (cmd == "tile" && argc >= 10) {
        built_scene b;
        if (!build(b, argv[2], iarg(11, 0), iarg(12, 0), iarg(13, 0), iarg(14, 0))) return usage();
        int x0 = iarg(3, 0), y0 = iarg(4, 0), w = iarg(5, 1), h = iarg(6, 1), spp = iarg(7, 0), threads = iarg(8, 1);
        std::string out = argv[9];
        bool ps = iarg(10, 0) != 0;
        ref_camera cam; cam.c = b.s.cam;
        if (spp <= 0) spp = cam.c.samples_per_pixel;
        std::vector<double> mean, per_sample; std::vector<uint32_t> counts;
        tile_result r = render_tile(b, cam, x0, y0, w, h, spp, threads, 1, 1, mean, ps ? &per_sample : nullptr, ps ? &counts : nullptr);
        write_npy(out + "_mean.npy", "<f8", {(size_t)h, (size_t)w, 3}, mean.data(), mean.size() * 8);
        if (ps) {
            write_npy(out + "_samples.npy", "<f8", {(size_t)h, (size_t)w, (size_t)spp, 3}, per_sample.data(), per_sample.size() * 8);
            write_npy(out + "_counts.npy", "<u4", {(size_t)h, (size_t)w, (size_t)spp, 2}, counts.data(), counts.size() * 4);
        }
        std::printf("{\"scene\": \"%s\", \"x0\": %d, \"y0\": %d, \"w\": %d, \"h\": %d, \"spp\": %d, \"max_depth\": %d, "
                    "\"image_width\": %d, \"image_height\": %d, \"seed\": %llu, \"segments\": %llu, \"draws\": %llu, "
                    "\"medium_draws\": %llu, \"objects\": %zu, \"bvh_build_s\": %.3f, \"render_s\": %.3f}\n",
                    argv[2], x0, y0, w, h, spp, cam.c.max_depth, cam.c.image_width, cam.c.image_height,
                    (unsigned long long)b.s.seed, (unsigned long long)r.segments, (unsigned long long)r.draws,
                    (unsigned long long)r.medium_draws, b.s.world.objects.size(), b.build_s, r.seconds);
        cleanup(b);
        return 0;
    }

    if (cmd == "aov" && argc >= 10) {
        // zenith_ref aov <scene> <x0> <y0> <w> <h> <zmax> <out_prefix> <unused> [a0..a3]: albedo / normal / z-depth tiles
        built_scene b;
        if (!build(b, argv[2], iarg(10, 0), iarg(11, 0), iarg(12, 0), iarg(13, 0))) return usage();
        int x0 = iarg(3, 0), y0 = iarg(4, 0), w = iarg(5, 1), h = iarg(6, 1);
        double zmax = std::atof(argv[7]);
        std::string out = argv[8];
        ref_camera cam; cam.c = b.s.cam; cam.setup();
        const int spp = cam.c.samples_per_pixel;
        const int aux_sample = std::clamp(spp / 8, 64, 1024);   // camera.hpp:433
        const int actual = std::min(aux_sample, spp);            // camera.hpp:535
        probe_world world(*b.bvh);
        std::vector<double> A((size_t)w * h * 3), N((size_t)w * h * 3), Z((size_t)w * h * 3);
        for (int jj = 0; jj < h; jj++)
            for (int ii = 0; ii < w; ii++) {
                int i = x0 + ii, j = y0 + jj;
                color a(0, 0, 0), n(0, 0, 0), z(0, 0, 0);
                for (int s2 = 0; s2 < spp; s2++) {
                    if (!(s2 < aux_sample)) break;
                    zr_oracle_seed(b.s.seed, (uint64_t)j * cam.c.image_width + i, (uint64_t)s2);
                    probe_world::sample_segments() = 0;
                    cam.aov_sample(i, j, world, zmax, a, n, z);
                }
                double sc = 1.0 / actual;
                color am = a * sc, nm = n * sc, zm = z * sc;
                size_t o = ((size_t)jj * w + ii) * 3;
                A[o] = am.x(); A[o + 1] = am.y(); A[o + 2] = am.z();
                N[o] = nm.x(); N[o + 1] = nm.y(); N[o + 2] = nm.z();
                Z[o] = zm.x(); Z[o + 1] = zm.y(); Z[o + 2] = zm.z();
            }
        write_npy(out + "_albedo.npy", "<f8", {(size_t)h, (size_t)w, 3}, A.data(), A.size() * 8);
        write_npy(out + "_normal.npy", "<f8", {(size_t)h, (size_t)w, 3}, N.data(), N.size() * 8);
        write_npy(out + "_zdepth.npy", "<f8", {(size_t)h, (size_t)w, 3}, Z.data(), Z.size() * 8);
        std::printf("{\"scene\": \"%s\", \"x0\": %d, \"y0\": %d, \"w\": %d, \"h\": %d, \"spp\": %d, \"aux\": %d, \"zmax\": %.17g, \"seed\": %llu}\n",
                    argv[2], x0, y0, w, h, spp, actual, zmax, (unsigned long long)b.s.seed);
        cleanup(b);
        return 0;
    }

    if (cmd == "passes" && argc >= 9) {
        // zenith_ref passes <scene> <x0> <y0> <w> <h> <spp|0> <out_prefix> [a0..a3]: beauty / reflection / refraction
        // tiles with use_reflection = use_refraction = true (camera.hpp:490-517, 531-533)
        built_scene b;
        if (!build(b, argv[2], iarg(9, 0), iarg(10, 0), iarg(11, 0), iarg(12, 0))) return usage();
        int x0 = iarg(3, 0), y0 = iarg(4, 0), w = iarg(5, 1), h = iarg(6, 1), spp = iarg(7, 0);
        std::string out = argv[8];
        ref_camera cam; cam.c = b.s.cam;
        if (spp > 0) cam.c.samples_per_pixel = spp;
        cam.setup();
        spp = cam.c.samples_per_pixel;
        probe_world world(*b.bvh);
        std::vector<double> B((size_t)w * h * 3), R((size_t)w * h * 3), F((size_t)w * h * 3);
        zr_oracle_tls.draws = 0; seg_tls.segments = 0;
        for (int jj = 0; jj < h; jj++)
            for (int ii = 0; ii < w; ii++) {
                int i = x0 + ii, j = y0 + jj;
                color cb(0, 0, 0), cr(0, 0, 0), cf(0, 0, 0);
                for (int s2 = 0; s2 < spp; s2++) {
                    zr_oracle_seed(b.s.seed, (uint64_t)j * cam.c.image_width + i, (uint64_t)s2);
                    probe_world::sample_segments() = 0;
                    cam.passes_sample(i, j, world, b.s.env, cb, cr, cf);
                }
                double sc = 1.0 / spp;  // light_scale, camera.hpp:436-437
                color bm = cb * sc, rm = cr * sc, fm = cf * sc;
                size_t o = ((size_t)jj * w + ii) * 3;
                B[o] = bm.x(); B[o + 1] = bm.y(); B[o + 2] = bm.z();
                R[o] = rm.x(); R[o + 1] = rm.y(); R[o + 2] = rm.z();
                F[o] = fm.x(); F[o + 1] = fm.y(); F[o + 2] = fm.z();
            }
        write_npy(out + "_beauty.npy", "<f8", {(size_t)h, (size_t)w, 3}, B.data(), B.size() * 8);
        write_npy(out + "_reflection.npy", "<f8", {(size_t)h, (size_t)w, 3}, R.data(), R.size() * 8);
        write_npy(out + "_refraction.npy", "<f8", {(size_t)h, (size_t)w, 3}, F.data(), F.size() * 8);
        std::printf("{\"scene\": \"%s\", \"x0\": %d, \"y0\": %d, \"w\": %d, \"h\": %d, \"spp\": %d, \"seed\": %llu, \"segments\": %llu, \"draws\": %llu}\n",
                    argv[2], x0, y0, w, h, spp, (unsigned long long)b.s.seed, (unsigned long long)seg_tls.segments,
                    (unsigned long long)zr_oracle_tls.draws);
        cleanup(b);
        return 0;
    }

    if (cmd == "post" && argc >= 5) {
        // zenith_ref post <scene> <preset> <out_prefix> [spp]: renders the scene's full frame (the input of the post stack),
        // then runs the GENUINE post_processor / bloom_filter through the loop of camera::process_framebuffer_to_image
        // (camera.hpp:701-780, restated: camera.hpp is unbuildable here) and post_processor::analyze_framebuffer.
        built_scene b;
        if (!build(b, argv[2], 0, 0, 0, 0)) return usage();
        const int preset = iarg(3, 0);
        std::string out = argv[4];
        const int spp = iarg(5, 4);
        ref_camera cam; cam.c = b.s.cam;
        const int W = cam.c.image_width, H = cam.c.image_height;
        std::vector<double> mean;
        render_tile(b, cam, 0, 0, W, H, spp, (int)std::thread::hardware_concurrency(), 1, 1, mean, nullptr, nullptr);
        std::vector<color> buffer((size_t)W * H);
        for (size_t i = 0; i < buffer.size(); i++) buffer[i] = color(mean[3 * i], mean[3 * i + 1], mean[3 * i + 2]);
        post_processor pp;
        bool is_data_pass = false, apply_gamma = true;
        switch (preset) {   // parameter sets that reach every branch of process()
            case 0: break;                                                     // defaults: exposure 0.5, vignette 1
            case 1: pp.exposure = 1.25f; pp.use_bloom = true; pp.bloom_threshold = 0.8f; pp.bloom_intensity = 0.45f; pp.bloom_radius = 6;
                    pp.use_aces_tone_mapping = true; break;
            case 2: pp.exposure = 0.2f; pp.use_sharpening = true; pp.sharpen_amount = 0.25; pp.contrast = 1.3f; pp.saturation = 1.4f;
                    pp.hue_shift = 35.0f; pp.color_balance = vec3(1.1f, 0.95f, 0.9f); pp.vignette_intensity = 0.6f; break;
            case 3: pp.exposure = 0.8f; pp.use_bloom = true; pp.use_sharpening = true; pp.saturation = 0.3f; pp.hue_shift = -120.0f;
                    pp.use_aces_tone_mapping = true; pp.vignette_intensity = 0.0f; pp.contrast = 0.8f; break;
            case 4: pp.debug.luminance = true; pp.exposure = 0.0f; break;
            case 5: pp.debug.red = true; pp.debug.blue = true; break;
            case 6: is_data_pass = true; break;                                // data pass: clamp + gamma
            case 7: is_data_pass = true; apply_gamma = false; break;
            default: return usage();
        }
        // camera.hpp:707-733
        std::vector<color> work = buffer;
        double ev = std::pow(2.0, (double)pp.exposure);
        if (!is_data_pass && pp.use_bloom) {
            for (auto& pix : work) pix *= ev;
            std::vector<color> glow(work.size(), color(0.0, 0.0, 0.0));
            bloom_filter bloom(pp.bloom_threshold, pp.bloom_intensity, pp.bloom_radius);
            bloom.generate_bloom_overlay(work, glow, W, H, 1.0f);
            double back = 1.0 / ev;
            for (size_t i = 0; i < work.size(); ++i) work[i] = buffer[i] + (glow[i] * back);
        } else {
            work = buffer;
        }
        if (!is_data_pass && pp.use_sharpening) pp.apply_sharpening(work, W, H, pp.sharpen_amount);
        std::vector<unsigned char> bytes((size_t)W * H * 3);
        for (int j = 0; j < H; j++)
            for (int i = 0; i < W; i++) {
                size_t k = static_cast<size_t>(j) * W + i;
                color c3 = work[k];
                if (!is_data_pass) {
                    c3 *= ev;
                    float u = static_cast<float>(i) / (W - 1);
                    float v = static_cast<float>(j) / (H - 1);
                    c3 = pp.process(c3, u, v, render_pass::RGB);
                } else {
                    c3 = color(std::clamp(c3.x(), 0.0, 1.0), std::clamp(c3.y(), 0.0, 1.0), std::clamp(c3.z(), 0.0, 1.0));
                    if (apply_gamma) c3 = linear_to_gamma(c3);
                }
                size_t idx = k * 3;
                bytes[idx + 0] = static_cast<unsigned char>(255.999 * c3.x());
                bytes[idx + 1] = static_cast<unsigned char>(255.999 * c3.y());
                bytes[idx + 2] = static_cast<unsigned char>(255.999 * c3.z());
            }
        image_statistics st = pp.analyze_framebuffer(buffer);
        double auto_on; { post_processor q = pp; q.use_auto_exposure = true; q.exposure_compensation_stops = 0.5f; auto_on = q.apply_auto_exposure(st); }
        double auto_off = pp.apply_auto_exposure(st);
        write_npy(out + "_frame.npy", "<f8", {(size_t)H, (size_t)W, 3}, mean.data(), mean.size() * 8);
        write_npy(out + "_rgb8.npy", "|u1", {(size_t)H, (size_t)W, 3}, bytes.data(), bytes.size());
        std::vector<int32_t> hist(st.histogram, st.histogram + 256);
        write_npy(out + "_hist.npy", "<i4", {256}, hist.data(), hist.size() * 4);
        std::printf("{\"scene\": \"%s\", \"preset\": %d, \"w\": %d, \"h\": %d, \"spp\": %d, \"is_data_pass\": %d, \"apply_gamma\": %d, "
                    "\"exposure\": %.9g, \"saturation\": %.9g, \"contrast\": %.9g, \"hue_shift\": %.9g, \"vignette_intensity\": %.9g, "
                    "\"bloom_threshold\": %.9g, \"bloom_intensity\": %.9g, \"bloom_radius\": %d, \"color_balance\": [%.17g, %.17g, %.17g], "
                    "\"sharpen_amount\": %.17g, \"use_aces_tone_mapping\": %d, \"use_bloom\": %d, \"use_sharpening\": %d, "
                    "\"debug\": [%d, %d, %d, %d, %d], \"average_luminance\": %.9g, \"max_luminance\": %.9g, \"auto_exposure_on\": %.17g, "
                    "\"auto_exposure_off\": %.17g}\n",
                    argv[2], preset, W, H, spp, (int)is_data_pass, (int)apply_gamma, pp.exposure, pp.saturation, pp.contrast, pp.hue_shift,
                    pp.vignette_intensity, pp.bloom_threshold, pp.bloom_intensity, pp.bloom_radius, pp.color_balance.x(), pp.color_balance.y(),
                    pp.color_balance.z(), pp.sharpen_amount, (int)pp.use_aces_tone_mapping, (int)pp.use_bloom, (int)pp.use_sharpening,
                    (int)pp.debug.red, (int)pp.debug.green, (int)pp.debug.blue, (int)pp.debug.luminance, (int)pp.debug.bvh,
                    st.average_luminance, st.max_luminance, auto_on, auto_off);
        cleanup(b);
        return 0;
    }

    if (cmd == "time" && argc >= 7) {
        // CPU baseline: every xstep-th column and ystep-th row of the full frame at `spp` samples
        built_scene b;
        if (!build(b, argv[2], iarg(7, 0), iarg(8, 0), iarg(9, 0), iarg(10, 0))) return usage();
        int xstep = std::max(1, iarg(3, 1)), ystep = std::max(1, iarg(4, 1)), spp = iarg(5, 0), threads = std::max(1, iarg(6, 1));
        ref_camera cam; cam.c = b.s.cam;
        if (spp <= 0) spp = cam.c.samples_per_pixel;
        int w = (cam.c.image_width + xstep - 1) / xstep, h = (cam.c.image_height + ystep - 1) / ystep;
        std::vector<double> mean;
        tile_result r = render_tile(b, cam, 0, 0, w, h, spp, threads, xstep, ystep, mean, nullptr, nullptr);
        double sum = 0; for (double v : mean) sum += v;
        std::printf("{\"scene\": \"%s\", \"pixels\": %d, \"spp\": %d, \"threads\": %d, \"segments\": %llu, \"primary\": %llu, "
                    "\"render_s\": %.4f, \"mseg_per_s\": %.4f, \"bvh_build_s\": %.3f, \"objects\": %zu, \"checksum\": %.17g}\n",
                    argv[2], w * h, spp, threads, (unsigned long long)r.segments, (unsigned long long)((uint64_t)w * h * spp),
                    r.seconds, r.segments / r.seconds * 1e-6, b.build_s, b.s.world.objects.size(), sum);
        cleanup(b);
        return 0;
    }

    if (cmd == "trace" && argc >= 6) {
        // known answers of world.hit on random rays: rays aimed from a shell around the scene box
        // towards points inside it, plus rays starting inside.  Record = hit_record fields.
        built_scene b;
        if (!build(b, argv[2], iarg(8, 0), iarg(9, 0), iarg(10, 0), iarg(11, 0))) return usage();
        const double clo = argc > 6 ? std::atof(argv[6]) : -30.0, chi = argc > 7 ? std::atof(argv[7]) : 600.0;
        size_t n = (size_t)iarg(3, 1000);
        uint64_t seed = (uint64_t)iarg(4, 1);
        std::string out = argv[5];
        aabb box = b.bvh->bounding_box();
        auto clampf = [](double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); };
        // keep the probe region finite for scenes with a huge ground sphere
        double lo[3] = {clampf(box.x.min, clo, chi), clampf(box.y.min, clo, chi), clampf(box.z.min, clo, chi)};
        double hi[3] = {clampf(box.x.max, clo, chi), clampf(box.y.max, clo, chi), clampf(box.z.max, clo, chi)};
        std::vector<double> rays(n * 6), recs(n * 16);
        const bool adversarial = std::getenv("ZR_TRACE_ADVERSARIAL") != nullptr;
        // map material pointers to first-seen ordinal so fixtures can compare material identity
        std::map<const material*, int> mat_id;
        probe_world world(*b.bvh);
        for (size_t k = 0; k < n; k++) {
            zr_oracle_seed(seed, 0x7ACEull, k);
            double q[6]; for (double& v : q) v = random_double();
            double o[3], t[3];
            for (int c = 0; c < 3; c++) {
                double ext = hi[c] - lo[c];
                o[c] = lo[c] - 0.25 * ext + 1.5 * ext * q[c];
                t[c] = lo[c] + ext * q[3 + c];
            }
            double dd[3] = {t[0] - o[0], t[1] - o[1], t[2] - o[2]};
            if (adversarial) {
                // rays that stress the box arithmetic of a traversal: exactly-zero direction components (both signs of
                // zero), tiny and large direction scales, origins on round coordinates, far origins
                switch (k % 8) {
                    case 0: dd[0] = 0.0; break;
                    case 1: dd[1] = -0.0; break;
                    case 2: dd[2] = 0.0; dd[0] = -0.0; break;
                    case 3: for (double& v : dd) v *= 1e-12; break;
                    case 4: for (double& v : dd) v *= 64.0; break;
                    case 5: for (int c = 0; c < 3; c++) { o[c] = std::round(o[c] * 4.0) / 4.0; dd[c] = t[c] - o[c]; } break;
                    case 6: for (int c = 0; c < 3; c++) { o[c] = t[c] + (o[c] - t[c]) * 4096.0; dd[c] = t[c] - o[c]; } break;
                    default: break;
                }
            }
            ray r(point3(o[0], o[1], o[2]), vec3(dd[0], dd[1], dd[2]));
            hit_record rec;
            probe_world::sample_segments() = 0;  // bounce 0 for the medium key
            bool h = world.hit(r, interval(0.001, infinity), rec);
            double* rr = &rays[k * 6];
            rr[0] = o[0]; rr[1] = o[1]; rr[2] = o[2]; rr[3] = dd[0]; rr[4] = dd[1]; rr[5] = dd[2];
            double* e = &recs[k * 16];
            if (h) {
                int id;
                auto it = mat_id.find(rec.mat.get());
                if (it == mat_id.end()) { id = (int)mat_id.size(); mat_id[rec.mat.get()] = id; } else id = it->second;
                e[0] = 1; e[1] = rec.t; e[2] = rec.p.x(); e[3] = rec.p.y(); e[4] = rec.p.z();
                e[5] = rec.normal.x(); e[6] = rec.normal.y(); e[7] = rec.normal.z();
                e[8] = rec.front_face ? 1 : 0; e[9] = rec.u; e[10] = rec.v;
                e[11] = rec.tangent.x(); e[12] = rec.tangent.y(); e[13] = rec.tangent.z();
                e[14] = id;
                // one scatter with the stream positioned at draw 0: attenuation checksum
                ray sc; color att;
                zr_oracle_tls.k = 0;
                bool s = rec.mat->scatter(r, rec, att, sc);
                e[15] = s ? (att.x() + 2 * att.y() + 4 * att.z()) : -1.0;
            } else {
                for (int c = 0; c < 16; c++) e[c] = 0;
            }
        }
        write_npy(out + "_rays.npy", "<f8", {n, 6}, rays.data(), rays.size() * 8);
        write_npy(out + "_recs.npy", "<f8", {n, 16}, recs.data(), recs.size() * 8);
        std::printf("{\"scene\": \"%s\", \"rays\": %zu, \"seed\": %llu, \"stream_pixel\": %llu}\n", argv[2], n,
                    (unsigned long long)seed, 0x7ACEull);
        cleanup(b);
        return 0;
    }

    if (cmd == "kat" && argc >= 7) {
        // zenith_ref kat <scene> hits|tex|bg|cam <in.bin> <n> <out_prefix>: per-function known answers on HAND-PLACED inputs
        // (raw little-endian doubles in, .npy out); the callee is always the genuine reference function:
        //   hits  n x (o, d, tmin, tmax)        -> world.hit(r, interval(tmin, tmax), rec) through bvh_node, then rec.mat->emitted and
        //                                          rec.mat->scatter with the stream (seed, 0x7ACE, k) at draw 0
        //   tex   n x (index, u, v, p)          -> zr_demo_scene::kat_textures[index]->value(u, v, p)
        //   bg    n x (mode, bg rgb, intensity, yaw, tilt, roll, sun dir, sun colour, sun intensity, sun size, dir)
        //                                       -> camera::get_background_color (HDR_MAP uses the scene's own hdr_texture)
        //   cam   n x (i, j, sample)            -> camera::initialize + get_ray(i, j) on the stream (scene seed, j W + i, sample)
        built_scene b;
        if (!build(b, argv[2], 0, 0, 0, 0)) return usage();
        const std::string what = argv[3];
        const size_t n = (size_t)std::atoll(argv[5]);
        const std::string out = argv[6];
        const size_t width = what == "hits" ? 8 : what == "tex" ? 6 : what == "bg" ? 19 : what == "cam" ? 3 : 0;
        if (!width) return usage();
        std::vector<double> in(n * width);
        { FILE* f = std::fopen(argv[4], "rb"); if (!f || std::fread(in.data(), 8, in.size(), f) != in.size()) { std::fprintf(stderr, "cannot read %s\n", argv[4]); return 2; } std::fclose(f); }
        if (what == "hits") {
            std::vector<double> recs(n * 16, 0.0), scat(n * 14, 0.0);
            std::map<const material*, int> mat_id;
            probe_world world(*b.bvh);
            const uint64_t seed = 0x5EED3001ull;
            for (size_t k = 0; k < n; k++) {
                const double* q = &in[k * 8];
                ray r(point3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]));
                zr_oracle_seed(seed, 0x7ACEull, k);
                hit_record rec;
                probe_world::sample_segments() = 0;
                bool h = world.hit(r, interval(q[6], q[7]), rec);
                double* e = &recs[k * 16];
                if (!h) continue;
                int id;
                auto it = mat_id.find(rec.mat.get());
                if (it == mat_id.end()) { id = (int)mat_id.size(); mat_id[rec.mat.get()] = id; } else id = it->second;
                e[0] = 1; e[1] = rec.t; e[2] = rec.p.x(); e[3] = rec.p.y(); e[4] = rec.p.z();
                e[5] = rec.normal.x(); e[6] = rec.normal.y(); e[7] = rec.normal.z();
                e[8] = rec.front_face ? 1 : 0; e[9] = rec.u; e[10] = rec.v;
                e[11] = rec.tangent.x(); e[12] = rec.tangent.y(); e[13] = rec.tangent.z();
                e[14] = id;
                double* sc = &scat[k * 14];
                color em = rec.mat->emitted(rec.u, rec.v, rec.p);
                ray scattered; color att;
                zr_oracle_tls.k = 0;
                const uint64_t d0 = zr_oracle_tls.draws;
                bool ok = rec.mat->scatter(r, rec, att, scattered);
                sc[0] = ok ? 1 : 0;
                if (ok) {
                    sc[1] = att.x(); sc[2] = att.y(); sc[3] = att.z();
                    sc[4] = scattered.origin().x(); sc[5] = scattered.origin().y(); sc[6] = scattered.origin().z();
                    sc[7] = scattered.direction().x(); sc[8] = scattered.direction().y(); sc[9] = scattered.direction().z();
                }
                sc[10] = em.x(); sc[11] = em.y(); sc[12] = em.z();
                sc[13] = (double)(zr_oracle_tls.draws - d0);
            }
            write_npy(out + "_recs.npy", "<f8", {n, 16}, recs.data(), recs.size() * 8);
            write_npy(out + "_scat.npy", "<f8", {n, 14}, scat.data(), scat.size() * 8);
            std::printf("{\"scene\": \"%s\", \"what\": \"hits\", \"n\": %zu, \"seed\": %llu, \"stream_pixel\": %llu}\n", argv[2], n, (unsigned long long)seed, 0x7ACEull);
        } else if (what == "tex") {
            std::vector<double> rgb(n * 3);
            for (size_t k = 0; k < n; k++) {
                const double* q = &in[k * 6];
                const size_t idx = (size_t)q[0];
                if (idx >= b.s.kat_textures.size()) { std::fprintf(stderr, "texture index out of range\n"); return 2; }
                color c = b.s.kat_textures[idx]->value(q[1], q[2], point3(q[3], q[4], q[5]));
                rgb[k * 3] = c.x(); rgb[k * 3 + 1] = c.y(); rgb[k * 3 + 2] = c.z();
            }
            write_npy(out + "_rgb.npy", "<f8", {n, 3}, rgb.data(), rgb.size() * 8);
            std::printf("{\"scene\": \"%s\", \"what\": \"tex\", \"n\": %zu, \"textures\": %zu}\n", argv[2], n, b.s.kat_textures.size());
        } else if (what == "bg") {
            std::vector<double> rgb(n * 3);
            ref_camera cam; cam.c = b.s.cam; cam.setup();
            for (size_t k = 0; k < n; k++) {
                const double* q = &in[k * 19];
                EnvironmentSettings env;
                env._mode = (EnvironmentSettings::Mode)(int)q[0];
                env.background_color = color(q[1], q[2], q[3]); env.intensity = q[4];
                env.hdri_rotation = q[5]; env.hdri_tilt = q[6]; env.hdri_roll = q[7];
                env.sun_direction = vec3(q[8], q[9], q[10]); env.sun_color = color(q[11], q[12], q[13]);
                env.sun_intensity = q[14]; env.sun_size = q[15];
                if (env._mode == EnvironmentSettings::HDR_MAP) env.hdr_texture = b.s.env.hdr_texture;
                color c = cam.get_background_color(ray(point3(0, 0, 0), vec3(q[16], q[17], q[18])), env);
                rgb[k * 3] = c.x(); rgb[k * 3 + 1] = c.y(); rgb[k * 3 + 2] = c.z();
            }
            write_npy(out + "_rgb.npy", "<f8", {n, 3}, rgb.data(), rgb.size() * 8);
            std::printf("{\"scene\": \"%s\", \"what\": \"bg\", \"n\": %zu}\n", argv[2], n);
        } else {
            std::vector<double> rays(n * 7);
            ref_camera cam; cam.c = b.s.cam; cam.setup();
            for (size_t k = 0; k < n; k++) {
                const int i = (int)in[k * 3], j = (int)in[k * 3 + 1], smp = (int)in[k * 3 + 2];
                zr_oracle_seed(b.s.seed, (uint64_t)j * cam.image_width + i, (uint64_t)smp);
                const uint64_t d0 = zr_oracle_tls.draws;
                ray r = cam.get_ray(i, j);
                double* o = &rays[k * 7];
                o[0] = r.origin().x(); o[1] = r.origin().y(); o[2] = r.origin().z();
                o[3] = r.direction().x(); o[4] = r.direction().y(); o[5] = r.direction().z();
                o[6] = (double)(zr_oracle_tls.draws - d0);
            }
            write_npy(out + "_rays.npy", "<f8", {n, 7}, rays.data(), rays.size() * 8);
            std::printf("{\"scene\": \"%s\", \"what\": \"cam\", \"n\": %zu, \"seed\": %llu, \"image_width\": %d, \"image_height\": %d}\n", argv[2], n,
                        (unsigned long long)b.s.seed, cam.image_width, cam.image_height);
        }
        cleanup(b);
        return 0;
    }

    if (cmd == "texels" && argc >= 5) {
        // decoded texels of the synthetic HDRI as the reference's stb loader returns them, sampled
        // through image_texture::value at every texel centre (public API only)
        int w = iarg(2, 64), h = iarg(3, 32);
        std::string p = "/tmp/zr_texels_" + std::to_string((long)getpid()) + ".hdr";
        zr_write_synthetic_hdr(p, w, h);
        image_texture tex(p.c_str(), true);
        std::vector<float> out((size_t)w * h * 3);
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                color c = tex.value((i + 0.5) / w, (j + 0.5) / h, point3(0, 0, 0));
                out[((size_t)j * w + i) * 3 + 0] = (float)c.x();
                out[((size_t)j * w + i) * 3 + 1] = (float)c.y();
                out[((size_t)j * w + i) * 3 + 2] = (float)c.z();
            }
        write_npy(argv[4], "<f4", {(size_t)h, (size_t)w, 3}, out.data(), out.size() * 4);
        ::unlink(p.c_str());
        return 0;
    }
    return usage();
}
