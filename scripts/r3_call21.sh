R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c21; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many_groups or composite" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
echo "== cfg3 (columns: stream_shade)" | tee $O/ab.txt
ZR_TIMELOG_KIND=2 BENCH_STEPS=3 bash scripts/ab_flags.sh "-DZR_SHADE_NO_PARTITION" 2>&1 | tee -a $O/ab.txt
echo "== cfg2 (columns: stream_shade)" | tee -a $O/ab.txt
ZR_TIMELOG_KIND=2 BENCH_STEPS=3 BENCH_ARGS="--workload cfg2" bash scripts/ab_flags.sh "-DZR_SHADE_NO_PARTITION" 2>&1 | tee -a $O/ab.txt
