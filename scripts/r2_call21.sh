R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2s
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 &&
BENCH_ARGS="--workload cfg3" timeout -k 10 900 bash scripts/ab_flags.sh "-DST_EXT_WAVES_LEAN=5" "-DST_EXT_WAVES_LEAN=4" "-DZR_NODE_WIDTH=4" 2>&1 | tee gpurun_out/r2s/width8.txt
for w in cfg2 cfg5 demo; do BENCH_ARGS="--workload $w" timeout -k 10 300 bash scripts/ab_env.sh 2>&1 | sed "s/^/$w /"; done | tee -a gpurun_out/r2s/width8.txt
