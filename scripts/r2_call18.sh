R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2q
BENCH_ARGS="--workload cfg3" timeout -k 10 900 bash scripts/ab_flags.sh "-DZR_EXT_WALK -DST_EXT_WAVES_LEAN=5" "-DST_EXT_WAVES_LEAN=5" "-DZR_EXT_WALK -DST_EXT_WAVES_LEAN=4" 2>&1 | tee gpurun_out/r2q/walk5.txt
