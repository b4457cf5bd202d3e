# round 3, call 6: full suite again (after the context pool), cfg4 rehearsals under bench.py's self-launch, shard time by tile size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c6; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -6 $O/pytest.txt
S() { python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print("ms_per_step", d["ms_per_step"], "extend per step", d["roofline"]["kernel_ms_per_step"], "launches", d["roofline"]["launches_timed"]//d["steps"], "segments", d["config"]["segments_per_step"])'; }
{
echo "== a rank's 1/8 share of cfg3 by tile size (ranks 0, 3, 7: load balance)"
for t in 32 64 128; do for r in 0 3 7; do echo "tile $t rank $r: $(ZR_MULTI_TILE=$t ZR_BENCH_SHARD_OF=8 ZR_BENCH_SHARD_RANK=$r python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | S)"; done; done
echo "== pools / units per slot on the 1/8 share (tile 32, rank 0)"
for kv in ZR_STREAM_POOLS=1 ZR_STREAM_POOLS=2 ZR_STREAM_POOLS=3 "ZR_STREAM_UNITS_PER_SLOT=6" "ZR_STREAM_UNITS_PER_SLOT=12" "ZR_BVH_BUILD=host"; do echo "$kv: $(env $kv ZR_BENCH_SHARD_OF=8 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | S)"; done
} > $O/shards.txt 2>&1
cat $O/shards.txt
for n in 2 4 6; do
  ZR_BENCH_ONE_DEVICE=1 ZR_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline > $O/rehearsal_${n}ranks.json 2> $O/rehearsal_${n}ranks.err; echo "rehearsal $n ranks exit $?: $(tail -c 400 $O/rehearsal_${n}ranks.json | head -c 400)"
done
