# round 3, call 3: full GPU suite (both builders, fused small-scene kernel), cfg5 fused vs pipeline, builder table for cfg3 / cfg3w / demo
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c3; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -8 $O/pytest.txt
L() { python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; c=d["config"]; print("ms_per_step", d["ms_per_step"], "value", d["value"], "kernel ms/launch", r["kernel_ms"], "launches/step", r["launches_timed"]//d["steps"], "kernel per step", r["kernel_ms_per_step"], "checksum", c["frame_checksum"], "commit_s", c["bvh_build_upload_s"], "first", c.get("bvh_build_upload_first_s"), c.get("bvh_builder"), "pairs", c["bvh_pairs"], "depth", c["bvh_depth"], "stack", c["traversal_stack"])'; }
{
echo "== cfg5: fused kernel vs pipeline"
for kv in ZR_FUSED=1 ZR_FUSED=0; do echo "$kv: $(env $kv python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload cfg5 2>/dev/null | L)"; done
echo "== cfg1 / cfg2 (fused limit: cfg2 has 485 spheres -> pipeline)"
for w in cfg1 cfg2; do echo "$w: $(python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | L)"; done
echo "== builders: host SAH vs device PLOC (r = 16, 32)"
for w in cfg3 cfg3w demo; do for kv in ZR_BVH_BUILD=host ZR_BVH_BUILD=device "ZR_BVH_BUILD=device ZR_BVH_PLOC_RADIUS=32"; do echo "$w $kv: $(env $kv python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | L)"; done; done
echo "== cfg3 device commit phases (second commit of the process)"
ZR_BVH_BUILD=device ZR_COMMIT_STATS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 8 2>&1 | grep -E "commit|device build" 
} > $O/ab.txt 2>&1
cat $O/ab.txt
