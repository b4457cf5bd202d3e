# development aid (CPU only): AddressSanitizer + UBSan and ThreadSanitizer builds of (1) the drop-in's scene flattening — every test scene twice,
# the 1M-triangle cfg3 on all threads (include/zenith/zenith.hpp: flatten_triangle_run, templates, the flat-scene stash) — and (2) the host BVH builder
# (raytracer_project_amd/csrc/zr_bvh.cpp, tests/native/bvh_check.cpp) on 300 000 objects with 8 threads.  Sanitizers are not available on the GPU pool:
# the device side is covered by the parity suite.  Usage: bash scripts/dev/sanitize.sh   (prints the reports, if any)
R=$(cd "$(dirname "$0")/../.." && pwd); T=${TMPDIR:-/tmp}/zr_san_$$; mkdir -p $T
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
for san in address,undefined thread; do
  tag=$(echo $san | cut -c1-3)
  $CXX -std=c++20 -O1 -g -fsanitize=$san -fno-omit-frame-pointer -pthread -ffp-contract=off -I$R/include -o $T/scenes_$tag $R/scripts/dev/sanitize_scenes.cpp \
      $R/raytracer_project_amd/csrc/zr_scenes_lib.cpp -L$R/raytracer_project_amd/csrc -lzr_hip -Wl,-rpath,$R/raytracer_project_amd/csrc 2>/dev/null || { echo "scenes ($san): build failed"; continue; }
  g++ -std=c++20 -O1 -g -fsanitize=$san -fno-omit-frame-pointer -pthread -I$R/raytracer_project_amd/csrc -o $T/bvh_$tag $R/tests/native/bvh_check.cpp $R/raytracer_project_amd/csrc/zr_bvh.cpp || { echo "bvh ($san): build failed"; continue; }
  echo "== scenes, -fsanitize=$san"; ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS=halt_on_error=0 $T/scenes_$tag 2>&1 | grep -v "^Model\|ok (\|Could not load texture" | head -40
  for args in "0 7 0" "5 3 2" "5000 11 3" "70000 5 1" "300000 9 0"; do echo "== bvh $args, -fsanitize=$san"; ZR_BVH_THREADS=8 TSAN_OPTIONS=halt_on_error=0 $T/bvh_$tag $args 2>&1 | grep -v '^{"n"' | head -20; done
done
rm -rf $T
echo "sanitize: done"
