# round 3, call 5: evidence after the builder / fused-kernel work — full suite, profile_round for cfg3 and cfg5, the bound counters (fixed script), shard timings
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -6 $O/pytest.txt
bash scripts/profile_round.sh r3 cfg3 > $O/profile_cfg3.txt 2>&1; tail -3 $O/profile_cfg3.txt
bash scripts/profile_round.sh r3 cfg5 > $O/profile_cfg5.txt 2>&1; tail -3 $O/profile_cfg5.txt
bash scripts/pmc_bound.sh $O/pmc_bound_cfg3 64 cfg3 > $O/pmc_bound_cfg3.txt 2>&1; tail -40 $O/pmc_bound_cfg3.txt
{
echo "== a rank's share of cfg3 (ZR_BENCH_SHARD_OF), default build"
for n in 8 4 2; do echo "1/$n: $(ZR_BENCH_SHARD_OF=$n python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print("ms_per_step", d["ms_per_step"], "extend per step", d["roofline"]["kernel_ms_per_step"], "launches", d["roofline"]["launches_timed"]//d["steps"], d["config"]["bvh_builder"], "commit", d["config"]["bvh_build_upload_s"])')"; done
echo "whole: $(python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print("ms_per_step", d["ms_per_step"], "extend per step", d["roofline"]["kernel_ms_per_step"], d["config"]["bvh_builder"], "commit", d["config"]["bvh_build_upload_s"])')"
} > $O/shards.txt 2>&1
cat $O/shards.txt
