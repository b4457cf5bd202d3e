R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c13; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -8 $O/pytest.txt
