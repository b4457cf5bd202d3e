# round 3, call 2: first light of the device BVH build — fixtures through both builders, then cfg3 timings per builder
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c2; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hit_records or radiance_matches_reference" > $O/pytest_fixtures.txt 2>&1; echo "fixtures exit $?" | tee -a $O/pytest_fixtures.txt
tail -25 $O/pytest_fixtures.txt
timeout -k 10 600 python -m pytest tests/test_fuzz_scenes.py -m gpu -x -q > $O/pytest_fuzz.txt 2>&1; echo "fuzz exit $?" | tee -a $O/pytest_fuzz.txt
tail -15 $O/pytest_fuzz.txt
B() { env "$@" ZR_COMMIT_STATS=1 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err_last.txt | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "extend ms/launch", r["kernel_ms"], "launches/step", r["launches_timed"]//d["steps"], "extend per step", r["kernel_ms_per_step"], "checksum", d["config"]["frame_checksum"], "commit_s", d["config"]["bvh_build_upload_s"], "pairs", d["config"]["bvh_pairs"], "depth", d["config"]["bvh_depth"], "stack", d["config"]["traversal_stack"])'; grep -E "commit|device build" $O/err_last.txt | head -30; }
{
for kv in ZR_BVH_BUILD=host ZR_BVH_BUILD=device "ZR_BVH_BUILD=device ZR_BVH_PLOC_RADIUS=8" "ZR_BVH_BUILD=device ZR_BVH_PLOC_RADIUS=32"; do echo "== cfg3 $kv"; B $kv; done
} > $O/ab.txt 2>&1
cat $O/ab.txt
