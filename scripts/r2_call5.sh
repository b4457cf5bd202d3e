R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2e
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2e/smoke.txt
grep -q "smoke OK" gpurun_out/r2e/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
nproc | tee gpurun_out/r2e/nproc.txt; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())" | tee -a gpurun_out/r2e/nproc.txt
ZR_COMMIT_STATS=1 ZR_BVH_PROFILE=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 2 > gpurun_out/r2e/cfg3.json 2> gpurun_out/r2e/cfg3.err; grep "\[zr\]" gpurun_out/r2e/cfg3.err | tee gpurun_out/r2e/commit_stats.txt; cut -c1-400 gpurun_out/r2e/cfg3.json
ZR_BENCH_SHARD_OF=8 BENCH_ARGS="" timeout -k 10 400 bash scripts/ab_flags.sh "-DST_CHUNK=128" "-DST_CHUNK=64" 2>&1 | tee gpurun_out/r2e/shard8_chunk.txt
ZR_BENCH_SHARD_OF=8 timeout -k 10 300 bash scripts/ab_env.sh "ZR_STREAM_POOLS=1" "ZR_STREAM_UNITS_PER_SLOT=4" "ZR_STREAM_UNITS_PER_SLOT=16" 2>&1 | tee gpurun_out/r2e/shard8_env.txt
bash scripts/r2_tests.sh
