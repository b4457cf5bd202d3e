// flatten_mixed_check.cpp — hittable_list::flatten of the drop-in (include/zenith/zenith.hpp) on long MIXED lists (the bulk path for runs of triangles
// must neither change what is flattened nor make a mixed list quadratic).  The list interleaves runs of triangles of many lengths — below and above the 16384
// that qualify a run for the bulk path, across its window sizes — with spheres and cubes; it is flattened twice, the second time with every triangle hidden
// behind a one-member hittable_list (which the bulk path cannot take: one by one), and the two flat scenes must hold the same primitives in the same order.
// Prints a JSON line {"n": .., "tris": .., "equal": .., "ms": ..}.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "zenith/zenith.hpp"

int main(int argc, char** argv) {
    const size_t target = argc > 1 ? (size_t)std::atoll(argv[1]) : 200000;
    const int pattern = argc > 2 ? std::atoi(argv[2]) : 0;
    shared_ptr<material> m0 = make_shared<lambertian>(color(0.5, 0.5, 0.5));
    shared_ptr<material> m1 = make_shared<metal>(color(0.8, 0.7, 0.6), 0.1);
    hittable_list bulk, plain;
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
    size_t n = 0, tris = 0;
    const size_t runs[] = {1, 3, 100, 16383, 16384, 16385, 40000, 65536, 65537, 5, 70000, 2, 300000, 7};
    size_t ri = 0;
    while (n < target) {
        size_t run = pattern == 1 ? 1 + (size_t)(rnd() * 40) : runs[ri++ % (sizeof runs / sizeof runs[0])];   // pattern 1: short runs only (the quadratic case)
        if (run > target - n) run = target - n;
        for (size_t k = 0; k < run; k++) {
            point3 a(rnd(), rnd(), rnd()), b(rnd(), rnd(), rnd()), c(rnd(), rnd(), rnd());
            vec3 nn(rnd(), rnd(), rnd());
            auto t = make_shared<triangle>(a, b, c, nn, nn, nn, (k % 7) ? m0 : m1);
            bulk.add(t);
            auto wrap = make_shared<hittable_list>(); wrap->add(t); plain.add(wrap);
        }
        n += run; tris += run;
        if (n < target) {
            shared_ptr<hittable> o;
            if ((ri + n) % 2) o = make_shared<sphere>(point3(rnd(), rnd(), rnd()), 0.1 + rnd(), m1);
            else o = make_shared<cube>(point3(-1, -1, -1), point3(1, 1, 1), m0);
            bulk.add(o); plain.add(o); n++;
        }
    }
    zenith::flat_scene fa, fb;
    const auto t0 = std::chrono::steady_clock::now();
    { zenith::scene_builder b(fa); bulk.flatten(b); b.finish(); }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    { zenith::scene_builder b(fb); plain.flatten(b); b.finish(); }
    bool eq = fa.tri_mat.size() == fb.tri_mat.size() && fa.sphere_mat == fb.sphere_mat && fa.cube_mat == fb.cube_mat && fa.spheres == fb.spheres && fa.cubes == fb.cubes &&
              fa.objects.size() == fb.objects.size() && fa.materials.size() == fb.materials.size();
    if (eq) eq = std::memcmp(fa.tri_v.data(), fb.tri_v.data(), fa.tri_v.size() * sizeof(double)) == 0 && std::memcmp(fa.tri_n.data(), fb.tri_n.data(), fa.tri_n.size() * sizeof(double)) == 0 &&
                 std::memcmp(fa.tri_mat.data(), fb.tri_mat.data(), fa.tri_mat.size() * sizeof(uint32_t)) == 0;
    if (eq) for (size_t k = 0; k < fa.objects.size(); k++) if (fa.objects[k].type != fb.objects[k].type || fa.objects[k].index != fb.objects[k].index) { eq = false; break; }
    std::printf("{\"n\": %zu, \"tris\": %zu, \"flat_tris\": %zu, \"equal\": %s, \"ms\": %.2f}\n", n, tris, (size_t)fa.tri_mat.size(), eq ? "true" : "false", ms);
    return eq && fa.tri_mat.size() == tris ? 0 : 1;
}
