# development aid: bench the in-tree build against the library built from another copy of the sources (a directory holding raytracer_project_amd/csrc, include, scenes,
# e.g. `git archive <rev> raytracer_project_amd/csrc include scenes | tar -x -C scripts/calls/base`), alternating, on the same box
#   BENCH_ARGS="--workload cfg3" BENCH_STEPS=3 REPS=2 bash scripts/ab_tree.sh scripts/calls/base [more trees ...]
R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --steps ${BENCH_STEPS:-3} --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "kernel ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"], "launches", r["launches_timed"], "checksum", d["config"]["frame_checksum"])'; }
for t in "$@"; do
  S=${TMPDIR:-/tmp}/zr_abt_$$_$(basename $t)
  mkdir -p $S && cp -r $R/$t/. $S/ && make -s -j8 -C $S/raytracer_project_amd/csrc > $S/build.log 2>&1 || { echo "$t: BUILD FAILED"; tail -5 $S/build.log; }
done
for rep in $(seq 1 ${REPS:-1}); do
  echo "in-tree: $(run)"
  for t in "$@"; do
    S=${TMPDIR:-/tmp}/zr_abt_$$_$(basename $t)
    echo "$t: $(ZR_LIB=$S/raytracer_project_amd/csrc/libzr_hip.so run)"
  done
done
