R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c26; mkdir -p $O
cd $R
run() { python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms", d["ms_per_step"], "extend", r["kernel_ms"], "records/launch", r["random_records_per_launch"], "alg bytes/launch", r["algorithmic_bytes_per_launch"], "util", r["lane_utilisation"], "launches", r["launches_timed"])'; }
for w in cfg3; do
export BENCH_ARGS="--workload $w"
echo "$w lists on: $(run)" | tee -a $O/ab.txt
echo "$w lists off: $(ZR_STREAM_LISTS=0 run)" | tee -a $O/ab.txt
done
python3 - <<'PY' 2>&1 | grep -v "^\[" | tee -a $O/ab.txt
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from raytracer_project_amd import capi
ds = capi.DemoScene("cfg3")
for lists in ("1", "0"):
    os.environ["ZR_STREAM_LISTS"] = lists
    c = capi.Context(0); sc = capi.Scene(c, ds.desc)
    cam = ds.camera.copy(); cam.samples_per_pixel = 16
    out = sc.render(cam, ds.env, ds.seed, None, count=True)
    ctr = sc.counters() if hasattr(sc, "counters") else c.counters()
    print("lists", lists, {k: getattr(ctr, k) for k in ("segments", "node_tests", "tri_tests", "sphere_tests") if hasattr(ctr, k)}, [f for f in dir(ctr) if not f.startswith("_")][:20])
    sc.close(); c.close()
PY
