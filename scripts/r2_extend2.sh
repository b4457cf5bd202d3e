# round 2, experiment 2: SHADE deals EXTEND's order (camera rays / direction octants) — smoke, A/B, wave profile, then the test suite
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2b
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2b/smoke.txt
grep -q "smoke OK" gpurun_out/r2b/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
BENCH_ARGS="--workload cfg3" timeout -k 10 500 bash scripts/ab_flags.sh "-DST_EXT_SORT=0" "-DST_EXT_SORT=0 -DZR_SHADE_HITMISS_ONLY" "-DST_EXT_SORT=0 -DST_CHUNK=128" 2>&1 | tee gpurun_out/r2b/ab_cfg3.txt
timeout -k 10 200 bash scripts/wave_profile.sh --full-only 2>&1 | tee gpurun_out/r2b/wp.txt
BENCH_ARGS="--workload cfg5" timeout -k 10 300 bash scripts/ab_flags.sh "-DST_EXT_SORT=0" 2>&1 | tee gpurun_out/r2b/ab_cfg5.txt
BENCH_ARGS="--workload cfg2" timeout -k 10 200 bash scripts/ab_flags.sh "-DST_EXT_SORT=0" 2>&1 | tee gpurun_out/r2b/ab_cfg2.txt
timeout -k 10 200 python3 bench.py --workload demo --no-cpu-baseline --steps 2 > gpurun_out/r2b/demo.json 2> gpurun_out/r2b/demo.err; tail -3 gpurun_out/r2b/demo.err; cut -c1-300 gpurun_out/r2b/demo.json
ZR_BENCH_SHARD_OF=8 timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('shard 1/8 ms_per_step', d['ms_per_step'])" | tee gpurun_out/r2b/shard8.txt
bash scripts/r2_tests.sh
