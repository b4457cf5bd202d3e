# round-2 evidence, part b: rocprofv3 stats + FETCH_SIZE / WRITE_SIZE passes (cfg3, cfg5), then the bench lines (which quote the traffic file of
# the same kernel sources), shard timings
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2g
bash scripts/profile_round.sh r2 cfg3 2>&1 | tail -3
bash scripts/profile_round.sh r2 cfg5 2>&1 | tail -3
cd $R
cp gpurun_out/r2_cfg3_traffic.json gpurun_out/r2_cfg5_traffic.json profiles/
python3 bench.py > gpurun_out/r2_cfg3_bench_with_cpu_baseline.json 2> gpurun_out/r2g/cfg3.err
for w in cfg2 cfg5 demo cfg3w; do python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/r2_${w}_bench.json 2> gpurun_out/r2g/$w.err; done
for n in 2 4 8; do ZR_BENCH_SHARD_OF=$n python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('shard 1/$n ms_per_step', d['ms_per_step'])"; done | tee gpurun_out/r2_shards.txt
for f in gpurun_out/r2_*_bench*.json; do echo "$f: $(cut -c1-230 $f)"; done
