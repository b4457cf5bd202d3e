R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2c
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2c/smoke.txt
grep -q "smoke OK" gpurun_out/r2c/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
bash scripts/pmc_bound.sh $R/gpurun_out/r2c/pmc_bound 64 cfg3 2>&1 | tee gpurun_out/r2c/pmc_bound.txt
BENCH_ARGS="--workload demo" timeout -k 10 300 bash scripts/ab_flags.sh "-DZR_SHADE_HITMISS_ONLY" 2>&1 | tee gpurun_out/r2c/ab_demo.txt
BENCH_ARGS="--workload cfg3" timeout -k 10 300 bash scripts/ab_flags.sh 2>&1 | tee gpurun_out/r2c/ab_cfg3.txt
bash scripts/r2_tests.sh
