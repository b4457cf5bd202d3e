R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2n
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 &&
for w in cfg3 cfg2 cfg5 demo; do BENCH_ARGS="--workload $w" timeout -k 10 300 bash scripts/ab_env.sh "ZR_TOP_LEVELS=0" "ZR_TOP_LEVELS=2" 2>&1 | sed "s/^/$w /"; done | tee gpurun_out/r2n/top_levels.txt
