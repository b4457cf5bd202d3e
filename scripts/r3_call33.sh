R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c33; mkdir -p $O
cd $R
echo "== cfg3 (stream_shade columns)" | tee $O/ab.txt
ZR_TIMELOG_KIND=2 BENCH_STEPS=3 bash scripts/ab_flags.sh "-DST_POOL_NT_ST" "-DST_SAMPLES_NT" "-DST_POOL_NT_ST -DST_SAMPLES_NT" 2>&1 | tee -a $O/ab.txt
