R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c10; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -12 $O/pytest.txt
