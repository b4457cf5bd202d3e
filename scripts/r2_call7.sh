R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2g
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2g/smoke.txt
grep -q "smoke OK" gpurun_out/r2g/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
ZR_COMMIT_STATS=1 ZR_BVH_PROFILE=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 8 2>&1 >/dev/null | grep "\[zr\]" | tee gpurun_out/r2g/commit_stats.txt
bash scripts/r2_tests.sh
grep -q "pytest exit 0" gpurun_out/r2t/pytest.txt || exit 1
bash scripts/profile_round.sh r2 cfg3 2>&1 | tail -3
bash scripts/profile_round.sh r2 cfg5 2>&1 | tail -3
cd $R
python3 bench.py > gpurun_out/r2_cfg3_bench_with_cpu_baseline.json 2> gpurun_out/r2g/cfg3.err
for w in cfg2 cfg5 demo cfg3w; do python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/r2_${w}_bench.json 2> gpurun_out/r2g/$w.err; done
for n in 2 4 8; do ZR_BENCH_SHARD_OF=$n python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('shard 1/$n ms_per_step', d['ms_per_step'])"; done | tee gpurun_out/r2_shards.txt
for f in gpurun_out/r2_*_bench*.json; do echo "$f: $(cut -c1-230 $f)"; done
