// zr_scenes_lib.cpp — libzr_scenes.so: the BASELINE.json scenes (scenes/zr_scenes.inc, the very file the
// oracle harness compiles against the genuine reference headers) compiled against the drop-in API
// (include/zenith/zenith.hpp).  Exposes (a) the flattened arrays, so that Python hosts (bench.py, tests)
// can feed them to the C ABI, and (b) a render through the drop-in's camera::render(world, env, post, flag),
// the same call the reference's main.cpp:1520-1521 makes.
#include <unistd.h>

#include <thread>

#include "../../include/zenith/zenith.hpp"

static void zr_hook_seed_scene(uint64_t seed, uint64_t stream) { zenith::seed_rng(seed, ZR_SCENE_PIXEL, stream); }
static shared_ptr<hittable> zr_hook_medium(shared_ptr<hittable> m) { return m; }  // media ids = flatten order

#include "../../scenes/zr_scenes.inc"
#include "../../scenes/zr_scenes_mix.inc"

namespace {
struct handle {
    zr_demo_scene s;
    zenith::flat_scene fs;
    zr_scene_desc desc{};
    zr_env env{};
    std::string warnings;
    std::vector<uint32_t> kat_tex;   // flattened ids of zr_demo_scene::kat_textures
};
}  // namespace

extern "C" {

void* zrs_build(const char* name, int a0, int a1, int a2, int a3) {
    handle* h = new handle();
    bool ok = zr_build_scene(name, h->s, a0, a1, a2, a3) || zr_build_scene_mix(name, h->s, a0);
    if (!ok) { delete h; return nullptr; }
    zenith::scene_builder b(h->fs);
    h->s.world.flatten(b);
    b.finish();
    h->env = zenith::to_zr_env(h->s.env, b);
    for (const auto& t : h->s.kat_textures) h->kat_tex.push_back(b.texture_id(t));
    h->desc = h->fs.desc();
    for (const auto& w : h->fs.warnings) { h->warnings += w; h->warnings += "\n"; }
    for (const auto& f : h->s.temp_files) ::unlink(f.c_str());
    h->s.temp_files.clear();
    return h;
}
void zrs_free(void* p) { delete (handle*)p; }
const zr_scene_desc* zrs_desc(void* p) { return &((handle*)p)->desc; }
const zr_camera* zrs_camera(void* p) { return &((handle*)p)->s.cam; }
const zr_env* zrs_env(void* p) { return &((handle*)p)->env; }
uint64_t zrs_seed(void* p) { return ((handle*)p)->s.seed; }
const char* zrs_warnings(void* p) { return ((handle*)p)->warnings.c_str(); }
int zrs_kat_textures(void* p, uint32_t* ids, int cap) {
    const handle* h = (const handle*)p;
    for (int k = 0; k < cap && k < (int)h->kat_tex.size(); k++) ids[k] = h->kat_tex[k];
    return (int)h->kat_tex.size();
}

// Renders through the drop-in C++ API exactly as the reference's caller does.  out = W*H*3 doubles.
// spp/width/height <= 0 keep the scene's values.  Returns 0, or -1 if the accumulator stayed empty.
int zrs_render_dropin(void* p, int width, int height, int spp, int device, double* out, zr_counters* ctr) {
    handle* h = (handle*)p;
    camera cam;
    const zr_camera& c = h->s.cam;
    cam.image_width = width > 0 ? width : c.image_width;
    cam.image_height = height > 0 ? height : c.image_height;
    cam.samples_per_pixel = spp > 0 ? spp : c.samples_per_pixel;
    cam.max_depth = c.max_depth; cam.vfov = c.vfov;
    cam.lookfrom = point3(c.lookfrom[0], c.lookfrom[1], c.lookfrom[2]);
    cam.lookat = point3(c.lookat[0], c.lookat[1], c.lookat[2]);
    cam.vup = vec3(c.vup[0], c.vup[1], c.vup[2]);
    cam.defocus_angle = c.defocus_angle; cam.focus_dist = c.focus_dist;
    cam.seed = h->s.seed; cam.device = device;
    cam.reset_accumulator();
    post_processor post;
    std::atomic<bool> flag{true};
    auto bvh_world = make_shared<bvh_node>(h->s.world);  // main.cpp:204
    cam.render(*bvh_world, h->s.env, post, flag);
    if (cam.lines_rendered.load() != cam.image_height) return -1;
    std::memcpy(out, cam.render_accumulator.data(), cam.render_accumulator.size() * sizeof(color));
    if (ctr) *ctr = cam.last_counters;
    return 0;
}

// The reference starts a fresh thread for every render (main.cpp:1520-1531): `n` renders of the scene, each on a thread of its own,
// one after the other.  Returns the number of device contexts the process has created so far (the drop-in keeps them in a
// process-wide pool: successive threads share one), or -1 if a render failed; `out` receives the last frame.
int zrs_render_dropin_threads(void* p, int width, int height, int spp, int device, int n, double* out) {
    int rc = 0;
    for (int k = 0; k < n && rc == 0; k++) {
        std::thread t([&]() { rc = zrs_render_dropin(p, width, height, spp, device, out, nullptr); });
        t.join();
    }
    return rc == 0 ? (int)zenith::contexts_created() : -1;
}

// The reference's whole frame pipeline through the drop-in API: render with auto-exposure and the reflection / refraction
// split, then the beauty image through the post stack (bloom + sharpening + ACES) and the reflection frame as a data pass —
// what main.cpp's render thread followed by save_render_pass(RGB) / save_render_pass(REFLECTIONS) does.
// rgb8 / refl8: W*H*3 bytes each; exposure_out: the auto-exposure value written back to post.exposure (camera.hpp:258-266)
int zrs_dropin_frame_to_rgb8(void* p, int spp, int device, unsigned char* rgb8, unsigned char* refl8, float* exposure_out) {
    handle* h = (handle*)p;
    camera cam;
    const zr_camera& c = h->s.cam;
    cam.image_width = c.image_width; cam.image_height = c.image_height;
    cam.samples_per_pixel = spp > 0 ? spp : c.samples_per_pixel;
    cam.max_depth = c.max_depth; cam.vfov = c.vfov;
    cam.lookfrom = point3(c.lookfrom[0], c.lookfrom[1], c.lookfrom[2]);
    cam.lookat = point3(c.lookat[0], c.lookat[1], c.lookat[2]);
    cam.vup = vec3(c.vup[0], c.vup[1], c.vup[2]);
    cam.defocus_angle = c.defocus_angle; cam.focus_dist = c.focus_dist;
    cam.seed = h->s.seed; cam.device = device;
    cam.use_reflection = true;
    cam.reset_accumulator();
    post_processor post;
    post.use_auto_exposure = true; post.exposure_compensation_stops = 0.5f;
    post.use_bloom = true; post.bloom_threshold = 0.8f; post.use_sharpening = true; post.use_aces_tone_mapping = true;
    std::atomic<bool> flag{true};
    auto bvh_world = make_shared<bvh_node>(h->s.world);
    cam.render(*bvh_world, h->s.env, post, flag);
    if (cam.lines_rendered.load() != cam.image_height) return -1;
    std::vector<unsigned char> a, b;
    if (!cam.process_framebuffer(cam.render_accumulator, post, a) || !cam.process_framebuffer(cam.reflection_buffer, post, b, true, true)) return -2;
    std::memcpy(rgb8, a.data(), a.size());
    std::memcpy(refl8, b.data(), b.size());
    if (exposure_out) *exposure_out = post.exposure;
    return 0;
}

// The reference's per-ray virtual API through the drop-in classes: for n rays (o, d, tmin, tmax) calls
// bvh_node(world).hit(r, interval(tmin, tmax), rec), then rec.mat->emitted(...) and rec.mat->scatter(r, rec, att, scattered)
// with random_double() positioned on the stream (seed, pixel, k) — the layout of `zenith_ref kat <scene> hits`:
// recs n x 16 (hit, t, p, normal, front_face, u, v, tangent, first-seen material ordinal, 0), scat n x 14 (scattered, attenuation,
// origin, direction, emitted, draws).  Returns 0, or -1 when a call threw.
int zrs_dropin_virtuals(void* p, const double* rays8, int n, uint64_t seed, uint64_t pixel, double* recs16, double* scat14) {
    handle* h = (handle*)p;
    try {
        auto world = make_shared<bvh_node>(h->s.world);
        std::unordered_map<const material*, int> ordinal;
        for (int k = 0; k < n; k++) {
            const double* q = rays8 + (size_t)k * 8;
            double* e = recs16 + (size_t)k * 16; double* sc = scat14 + (size_t)k * 14;
            for (int c = 0; c < 16; c++) e[c] = 0;
            for (int c = 0; c < 14; c++) sc[c] = 0;
            ray r(point3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]));
            zenith::seed_rng(seed, pixel, (uint64_t)k);
            hit_record rec;
            if (!world->hit(r, interval(q[6], q[7]), rec)) continue;
            zenith::seed_rng(seed, pixel, (uint64_t)k);
            e[0] = 1; e[1] = rec.t; e[2] = rec.p.x(); e[3] = rec.p.y(); e[4] = rec.p.z();
            e[5] = rec.normal.x(); e[6] = rec.normal.y(); e[7] = rec.normal.z(); e[8] = rec.front_face ? 1 : 0; e[9] = rec.u; e[10] = rec.v;
            e[11] = rec.tangent.x(); e[12] = rec.tangent.y(); e[13] = rec.tangent.z();
            if (!rec.mat) { e[14] = -1; continue; }
            auto it = ordinal.find(rec.mat.get());
            if (it == ordinal.end()) { const int id = (int)ordinal.size(); ordinal[rec.mat.get()] = id; e[14] = id; } else e[14] = it->second;
            color em = rec.mat->emitted(rec.u, rec.v, rec.p);
            ray scattered; color att;
            const uint64_t k0 = zenith::rng_state().k;
            bool ok = rec.mat->scatter(r, rec, att, scattered);
            sc[0] = ok ? 1 : 0;
            if (ok) {
                sc[1] = att.x(); sc[2] = att.y(); sc[3] = att.z();
                sc[4] = scattered.origin().x(); sc[5] = scattered.origin().y(); sc[6] = scattered.origin().z();
                sc[7] = scattered.direction().x(); sc[8] = scattered.direction().y(); sc[9] = scattered.direction().z();
            }
            sc[10] = em.x(); sc[11] = em.y(); sc[12] = em.z();
            sc[13] = (double)(zenith::rng_state().k - k0);
        }
    } catch (const std::exception& ex) {
        std::cerr << "[zrs_dropin_virtuals] " << ex.what() << "\n";
        return -1;
    }
    return 0;
}

}  // extern "C"

// sizes of the ABI structs as the C++ compiler sees them (tests compare them with the ctypes mirrors)
extern "C" size_t zrs_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(zr_xform_op);
        case 1: return sizeof(zr_object);
        case 2: return sizeof(zr_medium);
        case 3: return sizeof(zr_material);
        case 4: return sizeof(zr_texture);
        case 5: return sizeof(zr_env);
        case 6: return sizeof(zr_camera);
        case 7: return sizeof(zr_region);
        case 8: return sizeof(zr_counters);
        case 9: return sizeof(zr_hit);
        case 10: return sizeof(zr_scene_desc);
        case 11: return sizeof(zr_post_params);
        case 12: return sizeof(zr_image_stats);
        case 13: return sizeof(zr_aov_params);
        default: return 0;
    }
}
