# the GPU test suite, output kept under gpurun_out/
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2t
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2t/pytest.txt 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r2t/pytest.txt
tail -15 gpurun_out/r2t/pytest.txt
