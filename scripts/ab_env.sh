# development aid: bench the in-tree build under several environment settings ("VAR=value" each; "" = defaults)
R=$GRAFT_REPO_ROOT
for kv in "" "$@"; do
  echo "${kv:-default}: $(env $kv python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print("ms_per_step", d["ms_per_step"], "extend ms/launch", d["roofline"]["kernel_ms"], "per step", d["roofline"]["kernel_ms_per_step"], "checksum", d["config"]["frame_checksum"])')"
done
