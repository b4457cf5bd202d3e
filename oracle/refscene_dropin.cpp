// refscene_dropin.cpp — TEST INFRASTRUCTURE (oracle).  The reference's own scene code against the drop-in headers.
//
// zr_ref_scene_body.inc is cut out of /root/reference at build time by oracle/Makefile (into a scratch directory outside the repo,
// deleted after the compile): material_library.hpp without its #include lines, and scene_management.hpp:28-236 (sceneAssetsLoader,
// load_materials, build_geometry).  It is compiled here, UNCHANGED, against include/zenith/compat/*.hpp — the forwarding headers
// that carry the reference's file names — with ZENITH_STB_IMAGE so that image_texture decodes the reference's JPEG assets through
// the stb_image the reference vendors.  The resulting world is flattened by the drop-in (hittable::flatten) and rendered by the CPU
// restatement (libzr_oracle.so): tests/test_refdemo.py compares the tile with what the genuine reference renders from the same
// scene code (oracle/_ref/zenith_ref, scene "refdemo"), bit for bit.
//   refscene_dropin tile <x0> <y0> <w> <h> <spp> <out.npy>      radiance tile through the CPU restatement
//   refscene_dropin stats                                        object counts of the flattened world
//   refscene_dropin dump <prefix>                                the flattened world itself — every array of the zr_scene_desc, camera, environment, seed — as
//                                                                <prefix>_<name>.npy (structs as raw bytes): tests/golden/make_golden.py packs them into the
//                                                                fixture the GPU test commits through the C ABI (the scene's assets never travel; its arrays do)
#define ZENITH_STB_IMAGE
#define STB_IMAGE_IMPLEMENTATION
#include <unistd.h>
#include <map>
#include <string>
#include <vector>

#include "common.hpp"          // include/zenith/compat: every reference header name forwards to zenith.hpp

static void zr_hook_seed_scene(uint64_t seed, uint64_t stream) { zenith::seed_rng(seed, ZR_SCENE_PIXEL, stream); }
static shared_ptr<hittable> zr_hook_medium(shared_ptr<hittable> m) { return m; }

#include "zr_ref_scene_body.inc"
#include "../scenes/zr_scenes.inc"
#include "../scenes/zr_scene_refdemo.inc"

extern "C" {
void* zro_scene_create(const zr_scene_desc*);
void zro_scene_destroy(void*);
int zro_render(void*, const zr_camera*, const zr_env*, uint64_t, const zr_region*, int, double*, double*, uint32_t*, zr_counters*);
}

static void write_npy(const char* path, size_t h, size_t w, const double* data) {
    std::string hdr = "{'descr': '<f8', 'fortran_order': False, 'shape': (" + std::to_string(h) + "," + std::to_string(w) + ",3), }";
    size_t pad = (64 - (10 + hdr.size() + 1) % 64) % 64;
    hdr += std::string(pad, ' ') + "\n";
    FILE* f = std::fopen(path, "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path); std::exit(2); }
    unsigned char magic[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hdr.size() & 255), (unsigned char)(hdr.size() >> 8)};
    std::fwrite(magic, 1, 10, f); std::fwrite(hdr.data(), 1, hdr.size(), f); std::fwrite(data, 8, h * w * 3, f);
    std::fclose(f);
}

// a 1-D array as .npy: descr "<f8" / "<u4" / "|u1", n elements of `size` bytes
static void write_npy1(const std::string& path, const char* descr, size_t n, size_t size, const void* data) {
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (" + std::to_string(n) + ",), }";
    size_t pad = (64 - (10 + hdr.size() + 1) % 64) % 64;
    hdr += std::string(pad, ' ') + "\n";
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    unsigned char magic[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hdr.size() & 255), (unsigned char)(hdr.size() >> 8)};
    std::fwrite(magic, 1, 10, f); std::fwrite(hdr.data(), 1, hdr.size(), f);
    if (n) std::fwrite(data, size, n, f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: refscene_dropin tile <x0> <y0> <w> <h> <spp> <out.npy> | stats\n"); return 2; }
    zr_demo_scene s;
    zr_build_refdemo(s);
    zenith::flat_scene fs; zenith::scene_builder b(fs);
    s.world.flatten(b);
    b.finish();
    zr_env env = zenith::to_zr_env(s.env, b);
    zr_scene_desc d = fs.desc();
    if (std::string(argv[1]) == "stats") {
        std::printf("{\"objects\": %zu, \"triangles\": %zu, \"spheres\": %zu, \"cubes\": %zu, \"media\": %zu, \"materials\": %zu, \"textures\": %zu, \"texel_bytes\": %zu, \"warnings\": %zu}\n",
                    fs.objects.size(), fs.tri_mat.size(), fs.sphere_mat.size(), fs.cube_mat.size(), fs.media.size(), fs.materials.size(), fs.textures.size(), fs.texels.size(), fs.warnings.size());
        return 0;
    }
    if (std::string(argv[1]) == "dump") {
        if (argc < 3) return 2;
        const std::string pre = argv[2];
        auto f8 = [&](const char* name, const double* p, size_t n) { write_npy1(pre + "_" + name + ".npy", "<f8", n, 8, p); };
        auto u4 = [&](const char* name, const uint32_t* p, size_t n) { write_npy1(pre + "_" + name + ".npy", "<u4", n, 4, p); };
        auto raw = [&](const char* name, const void* p, size_t bytes) { write_npy1(pre + "_" + name + ".npy", "|u1", bytes, 1, p); };
        f8("spheres", d.spheres, d.n_spheres * 4); u4("sphere_mat", d.sphere_mat, d.n_spheres);
        f8("tri_v", d.tri_v, d.n_tris * 9); f8("tri_n", d.tri_n, d.n_tris * 9); u4("tri_mat", d.tri_mat, d.n_tris);
        f8("cubes", d.cubes, d.n_cubes * 12); u4("cube_mat", d.cube_mat, d.n_cubes);
        raw("media", d.media, d.n_media * sizeof(zr_medium)); raw("ops", d.ops, d.n_ops * sizeof(zr_xform_op)); raw("objects", d.objects, d.n_objects * sizeof(zr_object));
        raw("groups", d.groups, d.n_groups * sizeof(zr_group)); raw("materials", d.materials, d.n_materials * sizeof(zr_material));
        raw("textures", d.textures, d.n_textures * sizeof(zr_texture)); raw("texels", d.texels, d.texel_bytes);
        raw("camera", &s.cam, sizeof s.cam); raw("env", &env, sizeof env);
        std::printf("{\"scene\": \"refdemo\", \"seed\": %llu, \"objects\": %zu, \"triangles\": %zu, \"spheres\": %zu, \"cubes\": %zu, \"media\": %zu, \"ops\": %zu, \"groups\": %zu, "
                    "\"materials\": %zu, \"textures\": %zu, \"texel_bytes\": %zu}\n", (unsigned long long)s.seed, (size_t)d.n_objects, (size_t)d.n_tris, (size_t)d.n_spheres, (size_t)d.n_cubes,
                    (size_t)d.n_media, (size_t)d.n_ops, (size_t)d.n_groups, (size_t)d.n_materials, (size_t)d.n_textures, (size_t)d.texel_bytes);
        return 0;
    }
    if (argc < 8) return 2;
    zr_region reg{}; reg.x0 = std::atoi(argv[2]); reg.y0 = std::atoi(argv[3]); reg.w = std::atoi(argv[4]); reg.h = std::atoi(argv[5]);
    zr_camera cam = s.cam; cam.samples_per_pixel = std::atoi(argv[6]);
    std::vector<double> frame((size_t)cam.image_width * cam.image_height * 3, 0.0), tile((size_t)reg.w * reg.h * 3);
    void* sc = zro_scene_create(&d);
    zr_counters ctr{};
    if (!sc || zro_render(sc, &cam, &env, s.seed, &reg, 8, frame.data(), nullptr, nullptr, &ctr) != 0) { std::fprintf(stderr, "oracle render failed\n"); return 1; }
    for (int y = 0; y < reg.h; y++)
        std::memcpy(&tile[(size_t)y * reg.w * 3], &frame[((size_t)(reg.y0 + y) * cam.image_width + reg.x0) * 3], (size_t)reg.w * 3 * 8);
    write_npy(argv[7], reg.h, reg.w, tile.data());
    std::printf("{\"scene\": \"refdemo\", \"x0\": %d, \"y0\": %d, \"w\": %d, \"h\": %d, \"spp\": %d, \"segments\": %llu, \"draws\": %llu, \"objects\": %zu}\n",
                reg.x0, reg.y0, reg.w, reg.h, cam.samples_per_pixel, (unsigned long long)ctr.segments, (unsigned long long)ctr.rng_draws, fs.objects.size());
    zro_scene_destroy(sc);
    return 0;
}
