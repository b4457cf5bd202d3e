# round 3, call 7: fused kernel with its objects in the kernel arguments — small-scene tests, then cfg5 with build variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c7; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -6 $O/pytest.txt
BENCH_ARGS="--workload cfg5" bash scripts/ab_flags.sh "-DST_FUSED_WAVES=1" "-DST_FUSED_CHUNK=128" "-DST_FUSED_CHUNK=1024" > $O/ab_cfg5.txt 2>&1
cat $O/ab_cfg5.txt
