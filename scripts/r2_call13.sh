R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2m
BENCH_ARGS="--workload cfg3" timeout -k 10 600 bash scripts/ab_flags.sh "-DZR_EXT_PROBE=1" "-DZR_EXT_PROBE=2" 2>&1 | tee gpurun_out/r2m/ext_probe.txt
timeout -k 10 300 bash scripts/ab_env.sh "ZR_STREAM_POOLS=2" 2>&1 | tee -a gpurun_out/r2m/ext_probe.txt
