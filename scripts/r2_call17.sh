R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2p
for w in cfg5 demo; do BENCH_ARGS="--workload $w" timeout -k 10 400 bash scripts/ab_flags.sh "-DZR_SHADE_HITMISS_ONLY" 2>&1 | sed "s/^/$w /"; done | tee gpurun_out/r2p/hitmiss.txt
ZR_BENCH_SHARD_OF=8 timeout -k 10 300 bash scripts/ab_env.sh "ZR_STREAM_POOLS=1" 2>&1 | sed "s/^/shard8 /" | tee gpurun_out/r2p/shard.txt
for n in 4 2; do ZR_BENCH_SHARD_OF=$n timeout -k 10 300 bash scripts/ab_env.sh 2>&1 | sed "s/^/shard$n /" | tee -a gpurun_out/r2p/shard.txt; done
