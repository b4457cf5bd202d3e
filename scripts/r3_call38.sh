R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c38; mkdir -p $O
cd $R
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "value", d["value"], "extend ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"], "launches/step", r["launches_timed"]//d["steps"])'; }
for w in cfg2 demo; do
export BENCH_ARGS="--workload $w"
for u in 4 8 16 32 64; do echo "$w units/slot $u: $(ZR_STREAM_UNITS_PER_SLOT=$u run)" | tee -a $O/ab.txt; done
done
