# round 2, experiment 1: postponed-leaf EXTEND + material sort in SHADE — smoke, speed, lane utilisation, A/B
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2a
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2a/smoke.txt
grep -q "smoke OK" gpurun_out/r2a/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
BENCH_ARGS="--workload cfg3" timeout -k 10 500 bash scripts/ab_flags.sh "-DST_BIAS_NODE=1 -DST_BIAS_LEAF=2" "-DST_BIAS_NODE=3 -DST_BIAS_LEAF=2" "-DZR_SHADE_HITMISS_ONLY" 2>&1 | tee gpurun_out/r2a/ab_cfg3.txt
timeout -k 10 200 bash scripts/wave_profile.sh --full-only 2>&1 | tee gpurun_out/r2a/wp.txt
timeout -k 10 200 python3 scripts/counters.py cfg3 cfg5 cfg2 demo 2>&1 | tee gpurun_out/r2a/counters.txt
BENCH_ARGS="--workload cfg5" timeout -k 10 300 bash scripts/ab_flags.sh "-DZR_SHADE_HITMISS_ONLY" 2>&1 | tee gpurun_out/r2a/ab_cfg5.txt
BENCH_ARGS="--workload cfg2" timeout -k 10 200 bash scripts/ab_flags.sh 2>&1 | tee gpurun_out/r2a/ab_cfg2.txt
BENCH_ARGS="--workload demo" timeout -k 10 200 bash scripts/ab_flags.sh 2>&1 | tee gpurun_out/r2a/ab_demo.txt
