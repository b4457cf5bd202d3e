R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c20; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
BENCH_STEPS=3 bash scripts/ab_flags.sh "-DST_FETCH_MIN=8" "-DST_FETCH_MIN=24" "-DST_FETCH_MIN=32" "-DST_EXT_WAVES_LEAN=5" "-DST_EXT_WAVES_LEAN=7" "-DST_LDS_STACK=8" "-DST_LDS_STACK=16" "-DST_CHUNK=128" "-DST_CHUNK=512" 2>&1 | tee $O/ab.txt
