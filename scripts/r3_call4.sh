# round 3, call 4: SAH top over the PLOC clusters — fixtures through the device builder, then EXTEND per builder variant
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c4; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_scenes.py -m gpu -x -q -k "device" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
L() { python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; c=d["config"]; print("ms_per_step", d["ms_per_step"], "kernel ms/launch", r["kernel_ms"], "launches/step", r["launches_timed"]//d["steps"], "kernel per step", r["kernel_ms_per_step"], "checksum", c["frame_checksum"], "commit_s", c["bvh_build_upload_s"], "first", c.get("bvh_build_upload_first_s"), c.get("bvh_builder"), "pairs", c["bvh_pairs"], "depth", c["bvh_depth"], "stack", c["traversal_stack"])'; }
{
for w in cfg3 cfg3w; do for kv in ZR_BVH_BUILD=host "ZR_BVH_BUILD=device ZR_BVH_TOP=0" "ZR_BVH_BUILD=device ZR_BVH_TOP=1024" "ZR_BVH_BUILD=device ZR_BVH_TOP=4096" "ZR_BVH_BUILD=device ZR_BVH_TOP=16384" "ZR_BVH_BUILD=device ZR_BVH_TOP=65536"; do echo "$w $kv: $(env $kv python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | L)"; done; done
echo "== cfg3 device commit phases"
ZR_BVH_BUILD=device ZR_COMMIT_STATS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 8 2>&1 | grep -E "commit|device build" | tail -9
} > $O/ab.txt 2>&1
cat $O/ab.txt
