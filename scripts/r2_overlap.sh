# round 2, experiment 4: EXTEND || SHADE.  Two sub-pools on two streams, each pool's persistent EXTEND grid limited to a share of
# the wave slots so that the other pool's SHADE blocks can be resident beside it (ZR_ST_BLOCKS = EXTEND workgroups per launch;
# the full chip holds 256 CUs x 24 = 6144 at 6 waves/SIMD).
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2d
BENCH_ARGS="--workload cfg3" timeout -k 10 700 bash scripts/ab_env.sh "ZR_STREAM_POOLS=2" "ZR_STREAM_POOLS=2 ZR_ST_BLOCKS=4096" "ZR_STREAM_POOLS=2 ZR_ST_BLOCKS=3072" "ZR_STREAM_POOLS=2 ZR_ST_BLOCKS=2048" "ZR_STREAM_POOLS=3 ZR_ST_BLOCKS=2048" "ZR_STREAM_POOLS=4 ZR_ST_BLOCKS=1536" "ZR_ST_BLOCKS=3072" 2>&1 | tee gpurun_out/r2d/overlap_cfg3.txt
BENCH_ARGS="--workload cfg5" timeout -k 10 300 bash scripts/ab_env.sh "ZR_STREAM_POOLS=2 ZR_ST_BLOCKS=2048" "ZR_STREAM_POOLS=2 ZR_ST_BLOCKS=1536" 2>&1 | tee gpurun_out/r2d/overlap_cfg5.txt
BENCH_ARGS="--workload cfg2" timeout -k 10 200 bash scripts/ab_env.sh "ZR_STREAM_POOLS=2 ZR_ST_BLOCKS=3072" 2>&1 | tee gpurun_out/r2d/overlap_cfg2.txt
