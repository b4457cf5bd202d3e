# development aid: bench the in-tree build, then rebuild the library with extra definitions ("$@", one build each) and bench again
R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("ms_per_step", d["ms_per_step"], "extend ms/launch", d["roofline"]["kernel_ms"], "per step", d["roofline"]["kernel_ms_per_step"], "checksum", d["config"]["frame_checksum"])'; }
echo "in-tree: $(run)"
for f in "$@"; do
  touch $R/raytracer_project_amd/csrc/*.hip $R/raytracer_project_amd/csrc/*.cpp
  make -s -C $R/raytracer_project_amd/csrc ZR_DEFS="$f" > /dev/null 2>&1
  echo "$f: $(run)"
done
