# development aid: like ab_flags.sh, but the variants are compiler options for the kernels only (make ZR_KFLAGS="...")
R=$GRAFT_REPO_ROOT
S=${TMPDIR:-/tmp}/zr_abk_$$
run() { python3 $R/bench.py --steps ${BENCH_STEPS:-3} --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "extend ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"], "launches", r["launches_timed"], "checksum", d["config"]["frame_checksum"])'; }
echo "in-tree: $(run)"
mkdir -p $S/raytracer_project_amd && cp -r $R/include $S/ && cp -r $R/scenes $S/ && cp -r $R/raytracer_project_amd/csrc $S/raytracer_project_amd/
for f in "$@"; do
  touch $S/raytracer_project_amd/csrc/zr_stream.hip
  make -s -j8 -C $S/raytracer_project_amd/csrc ZR_KFLAGS="$f" > $S/build.log 2>&1 || { echo "$f: BUILD FAILED"; tail -3 $S/build.log; continue; }
  echo "$f: $(ZR_LIB=$S/raytracer_project_amd/csrc/libzr_hip.so run)"
done
rm -rf $S
