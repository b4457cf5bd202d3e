R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c23; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
echo "== cfg3" | tee $O/ab.txt
BENCH_STEPS=3 bash scripts/ab_flags.sh "-DST_NO_ROOT_LEAVES" 2>&1 | tee -a $O/ab.txt
echo "== cfg2" | tee -a $O/ab.txt
BENCH_STEPS=3 BENCH_ARGS="--workload cfg2" bash scripts/ab_flags.sh "-DST_NO_ROOT_LEAVES" 2>&1 | tee -a $O/ab.txt
