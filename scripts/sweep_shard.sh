# development aid: frame time on one GPU for 1/N shards and several sub-pool counts
R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["config"]["frame_checksum"])'; }
for pools in 1 2 3 4 6 8; do
  echo "shard 1/8 pools $pools: $(ZR_STREAM_POOLS=$pools ZR_BENCH_SHARD_OF=8 run)"
done
for pools in 4 8; do
  echo "shard 1/8 pools $pools slots 32M: $(ZR_STREAM_SLOTS=33554432 ZR_STREAM_UNITS_PER_SLOT=4 ZR_STREAM_POOLS=$pools ZR_BENCH_SHARD_OF=8 run)"
done
for pools in 1 2 4 8; do
  echo "full pools $pools: $(ZR_STREAM_POOLS=$pools run)"
done
