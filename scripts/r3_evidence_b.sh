# round 3 evidence, part B: bench lines per workload, the builder table, a rank's share, the N-rank rehearsals on one device
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3eb; mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/r3_cfg3_bench_with_cpu_baseline.json 2> $O/cfg3.err; tail -c 300 $O/r3_cfg3_bench_with_cpu_baseline.json; echo
for w in cfg2 cfg5 cfg3w demo cfg1; do python3 bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline > $O/r3_${w}_bench.json 2> $O/$w.err; done
L() { python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; c=d["config"]; print("ms_per_step", d["ms_per_step"], "value", d["value"], r["kernel"], "ms/launch", r["kernel_ms"], "launches/step", r["launches_timed"]//d["steps"], "kernel per step", r["kernel_ms_per_step"], "checksum", c["frame_checksum"], "commit_s", c["bvh_build_upload_s"], "first", c.get("bvh_build_upload_first_s"), c.get("bvh_builder"), "pairs", c["bvh_pairs"], "depth", c["bvh_depth"], "stack", c["traversal_stack"])'; }
{
echo "# host SAH builder (zr_bvh.cpp) against the device builder (zr_build.hip: PLOC + SAH top over ZR_BVH_TOP clusters), one MI355X, bench.py --steps 3 --warmup 1"
for w in cfg3 cfg3w demo; do for kv in ZR_BVH_BUILD=host "ZR_BVH_BUILD=device ZR_BVH_TOP=0" ZR_BVH_BUILD=device "ZR_BVH_BUILD=device ZR_BVH_TOP=65536"; do echo "$w $kv: $(env $kv python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | L)"; done; done
echo "# cfg3, device build: phases of the second commit of a process"
ZR_BVH_BUILD=device ZR_COMMIT_STATS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 8 2>&1 | grep -E "commit|device build" | tail -9
} > $O/r3_builders.txt 2>&1
cat $O/r3_builders.txt
{
echo "# a rank's share of cfg3 on one MI355X (ZR_BENCH_SHARD_OF=N), then the whole frame on the same box"
for n in 8 4 2; do echo "1/$n: $(ZR_BENCH_SHARD_OF=$n python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | L)"; done
echo "whole: $(python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | L)"
} > $O/r3_shards.txt 2>&1
cat $O/r3_shards.txt
for n in 2 4; do
  ZR_BENCH_ONE_DEVICE=1 ZR_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline > $O/r3_rehearsal_${n}ranks_one_gpu_gloo.json 2> $O/rehearsal_${n}.err; echo "rehearsal $n ranks exit $?"
done
