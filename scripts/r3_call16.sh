R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c16; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
BENCH_STEPS=3 bash scripts/ab_flags.sh "-DST_NODE_MINMAX" "-DST_SORT_PARTIAL" "-DST_RCP_FAST" 2>&1 | tee $O/ab.txt
