R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2q
BENCH_ARGS="--workload cfg3" timeout -k 10 900 bash scripts/ab_flags.sh "-DZR_TRI_STRIDE=16" "-DZR_TRI_STRIDE=10" 2>&1 | tee gpurun_out/r2q/tri_stride.txt
