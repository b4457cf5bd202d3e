R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c11; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep_tree" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
ZR_TIMELOG_KIND=2 bash scripts/ab_flags.sh "-DST_SHADE_WAVES=5" "-DST_SHADE_WAVES=6" "-DST_SHADE_WAVES=3" > $O/ab_shade.txt 2>&1
cat $O/ab_shade.txt
