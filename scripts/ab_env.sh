# development aid: bench the in-tree build under several environment settings ("VAR=value" each)
R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("ms_per_step", d["ms_per_step"], "extend ms/launch", d["roofline"]["kernel_ms"], "per step", d["roofline"]["kernel_ms_per_step"], "checksum", d["config"]["frame_checksum"])'; }
echo "default: $(run)"
for kv in "$@"; do echo "$kv: $(env $kv bash -c "$(declare -f run); R=$R run")"; done
