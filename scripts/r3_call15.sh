R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c15; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -8 $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
