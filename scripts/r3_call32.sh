R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c32; mkdir -p $O
cd $R
echo "== cfg3 (stream_extend columns)" | tee $O/ab.txt
BENCH_STEPS=3 bash scripts/ab_flags.sh "-DST_POOL_NT" 2>&1 | tee -a $O/ab.txt
echo "== cfg3 (stream_shade columns)" | tee -a $O/ab.txt
ZR_TIMELOG_KIND=2 BENCH_STEPS=3 bash scripts/ab_flags.sh "-DST_POOL_NT" 2>&1 | tee -a $O/ab.txt
echo "== cfg2" | tee -a $O/ab.txt
BENCH_STEPS=3 BENCH_ARGS="--workload cfg2" bash scripts/ab_flags.sh "-DST_POOL_NT" 2>&1 | tee -a $O/ab.txt
