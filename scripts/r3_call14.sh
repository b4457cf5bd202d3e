R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c14; mkdir -p $O
cd $R
BENCH_ARGS="--workload cfg5" bash scripts/ab_flags.sh "-DST_FUSED_LDS_STATE" > $O/ab_cfg5.txt 2>&1
cat $O/ab_cfg5.txt
