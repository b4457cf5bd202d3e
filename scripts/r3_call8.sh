R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c8; mkdir -p $O
cd $R
BENCH_ARGS="--workload cfg5" bash scripts/ab_flags.sh "-DST_FUSED_WAVES=3" "-DST_FUSED_WAVES=4" > $O/ab_cfg5.txt 2>&1
cat $O/ab_cfg5.txt
