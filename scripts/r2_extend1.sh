# round 2, experiment 1: postponed-leaf EXTEND — parity, then speed and lane utilisation, then the NODE:LEAF bias
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.txt 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r2a/pytest.txt
tail -3 gpurun_out/r2a/pytest.txt
grep -q " passed" gpurun_out/r2a/pytest.txt || exit 1
BENCH_ARGS="--workload cfg3" bash scripts/ab_flags.sh "-DST_BIAS_NODE=1 -DST_BIAS_LEAF=2" "-DST_BIAS_NODE=2 -DST_BIAS_LEAF=3" "-DST_BIAS_NODE=3 -DST_BIAS_LEAF=2" 2>&1 | tee gpurun_out/r2a/ab_cfg3.txt
bash scripts/wave_profile.sh --full-only 2>&1 | tee gpurun_out/r2a/wp.txt
python3 scripts/counters.py cfg3 2>&1 | tee gpurun_out/r2a/counters.txt
BENCH_ARGS="--workload cfg5" bash scripts/ab_flags.sh 2>&1 | tee gpurun_out/r2a/ab_cfg5.txt
BENCH_ARGS="--workload cfg2" bash scripts/ab_flags.sh 2>&1 | tee gpurun_out/r2a/ab_cfg2.txt
