R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c18; mkdir -p $O
cd $R
BENCH_STEPS=3 bash scripts/ab_flags.sh "-DST_DUMMY_VALU=24" "-DST_DUMMY_VALU=48" "-DST_DUMMY_VALU=96" 2>&1 | tee $O/ab.txt
