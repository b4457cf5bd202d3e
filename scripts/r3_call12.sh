R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c12; mkdir -p $O
cd $R
S() { python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print("ms_per_step", d["ms_per_step"], "extend per step", d["roofline"]["kernel_ms_per_step"], "launches", d["roofline"]["launches_timed"]//d["steps"])'; }
{
echo "== 1/8 share of cfg3: units per slot x pools"
for u in 3 4 5 8; do for k in 2; do echo "units/slot $u pools $k: $(ZR_STREAM_UNITS_PER_SLOT=$u ZR_STREAM_POOLS=$k ZR_BENCH_SHARD_OF=8 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | S)"; done; done
echo "units/slot 4 pools 1: $(ZR_STREAM_UNITS_PER_SLOT=4 ZR_STREAM_POOLS=1 ZR_BENCH_SHARD_OF=8 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | S)"
echo "units/slot 2 pools 2: $(ZR_STREAM_UNITS_PER_SLOT=2 ZR_STREAM_POOLS=2 ZR_BENCH_SHARD_OF=8 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | S)"
echo "== whole frame: pools 1 / 2"
for k in 1 2; do echo "pools $k: $(ZR_STREAM_POOLS=$k python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | S)"; done
} > $O/shard_sweep.txt 2>&1
cat $O/shard_sweep.txt
