R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2k
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "radiance_matches_reference or passes or full_size" 2>&1 | tail -3
for w in cfg3 cfg5 demo; do BENCH_ARGS="--workload $w" timeout -k 10 500 bash scripts/ab_flags.sh "-DZR_SHADE_NO_TOUCH" "-DZR_EXT_LAZY_RAY" 2>&1 | sed "s/^/$w /"; done | tee gpurun_out/r2k/early_loads.txt
