R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c30; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -6 $O/pytest.txt
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "value", d["value"], "extend ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"], "launches", r["launches_timed"], "checksum", d["config"]["frame_checksum"], r["kernel"])'; }
for w in demo cfg5 cfg3 cfg2 cfg5; do BENCH_ARGS="--workload $w"; echo "$w: $(run)" | tee -a $O/ab.txt; done
