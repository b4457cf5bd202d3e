#include <cstdio>
#include <cstdint>
extern "C" { void* zrs_build(const char*, int, int, int, int); void zrs_free(void*); const char* zrs_warnings(void*); }
struct zr_scene_desc_fwd;
extern "C" const void* zrs_desc(void*);
int main() {
    const char* names[] = {"cfg1", "cfg2", "cfg5", "mix0", "mix1", "mix2", "mesh0", "inst0", "inst1", "inst2", "demo", "kat0", "cfg3w", "cfg3"};
    for (const char* n : names) {
        for (int rep = 0; rep < 2; rep++) {
            void* h = zrs_build(n, 0, 0, 0, 0);
            if (!h) { std::printf("%s: build failed\n", n); return 1; }
            std::printf("%s ok (%s)\n", n, zrs_warnings(h)[0] ? "warnings" : "no warnings");
            zrs_free(h);
        }
    }
    return 0;
}
