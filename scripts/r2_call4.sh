R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2d
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2d/smoke.txt
grep -q "smoke OK" gpurun_out/r2d/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
bash scripts/r2_overlap.sh
BENCH_ARGS="--workload cfg5" timeout -k 10 400 bash scripts/ab_flags.sh "-DST_EXT_WAVES_MID=4" "-DST_EXT_WAVES_MID=6" 2>&1 | tee gpurun_out/r2d/ab_cfg5.txt
ZR_BAKE_TRIANGLES=0 timeout -k 10 200 python3 bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | cut -c1-160 | tee gpurun_out/r2d/cfg5_nobake.txt
timeout -k 10 200 python3 scripts/counters.py cfg5 demo 2>&1 | tee gpurun_out/r2d/counters.txt
BENCH_ARGS="--workload demo" timeout -k 10 300 bash scripts/ab_flags.sh "-DZR_SHADE_HITMISS_ONLY" 2>&1 | tee gpurun_out/r2d/ab_demo.txt
bash scripts/r2_tests.sh
