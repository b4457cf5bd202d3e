R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2j
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for w in cfg3 cfg2 cfg5 demo; do BENCH_ARGS="--workload $w" timeout -k 10 300 bash scripts/ab_env.sh "ZR_STREAM_BOTTOM_UP=0" 2>&1 | sed "s/^/$w /"; done | tee gpurun_out/r2j/bottom_up.txt
ZR_BENCH_SHARD_OF=8 timeout -k 10 300 bash scripts/ab_env.sh "ZR_STREAM_BOTTOM_UP=0" "ZR_STREAM_POOLS=1" 2>&1 | sed "s/^/shard8 /" | tee -a gpurun_out/r2j/bottom_up.txt
python3 scripts/rounds_shard.py 8 2>/dev/null | tee gpurun_out/r2j/rounds_shard8.txt
