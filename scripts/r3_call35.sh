R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c35; mkdir -p $O
cd $R
echo "== cfg3 (stream_shade columns)" | tee $O/ab.txt
ZR_TIMELOG_KIND=2 BENCH_STEPS=4 bash scripts/ab_flags.sh "-DST_KEY_DERIVE" 2>&1 | tee -a $O/ab.txt
ZR_TIMELOG_KIND=2 BENCH_STEPS=4 bash scripts/ab_flags.sh "-DST_KEY_DERIVE" 2>&1 | tee -a $O/ab.txt
