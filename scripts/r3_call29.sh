R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c29; mkdir -p $O
cd $R
run() { ZR_TIMELOG_KIND=2 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "shade ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"], "launches", r["launches_timed"], "checksum", d["config"]["frame_checksum"])'; }
for g in 0 1024 2048 4096 16384; do echo "ZR_SHADE_GRID=$g: $(ZR_SHADE_GRID=$g run)" | tee -a $O/ab.txt; done
