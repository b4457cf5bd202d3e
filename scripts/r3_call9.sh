R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c9; mkdir -p $O
cd $R
bash scripts/ab_flags.sh "-DST_LEAF_ALL" "-DST_LEAF_ALL -DST_BIAS_NODE=1 -DST_BIAS_LEAF=1" "-DST_LEAF_ALL -DST_BIAS_NODE=2 -DST_BIAS_LEAF=3" > $O/ab_cfg3.txt 2>&1
cat $O/ab_cfg3.txt
BENCH_ARGS="--workload cfg2" bash scripts/ab_flags.sh "-DST_LEAF_ALL" > $O/ab_cfg2.txt 2>&1
cat $O/ab_cfg2.txt
