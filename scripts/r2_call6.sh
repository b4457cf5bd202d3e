R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2f
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2f/smoke.txt
grep -q "smoke OK" gpurun_out/r2f/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
for t in 16 32 64; do echo "== ZR_BVH_THREADS=$t"; ZR_BVH_THREADS=$t ZR_COMMIT_STATS=1 ZR_BVH_PROFILE=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 8 2>&1 >/dev/null | grep "\[zr\]"; done | tee gpurun_out/r2f/commit_stats.txt
ZR_COMMIT_STATS=1 timeout -k 10 200 python3 bench.py --workload demo --no-cpu-baseline --steps 1 --warmup 0 --spp 8 2>&1 >/dev/null | grep "\[zr\]" | tee gpurun_out/r2f/commit_stats_demo.txt
bash scripts/r2_tests.sh
( time timeout -k 10 600 python3 bench.py > gpurun_out/r2f/bench_default.json 2> gpurun_out/r2f/bench_default.err ) 2>&1 | tail -3 | tee gpurun_out/r2f/bench_default_time.txt
cut -c1-300 gpurun_out/r2f/bench_default.json
