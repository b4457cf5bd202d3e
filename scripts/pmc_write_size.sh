# WRITE_SIZE of the pipeline kernels in its own --pmc pass (FETCH_SIZE + WRITE_SIZE do not fit one pass on gfx950)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcW
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcW_bench.json 2> $R/gpurun_out/pmcW.err
python3 - <<'PY'
import csv, glob, os, collections, json
R = os.environ['GRAFT_REPO_ROOT']
f = sorted(glob.glob(f'{R}/gpurun_out/pmcW/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
agg = collections.defaultdict(float); seen = set(); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != 'WRITE_SIZE': continue
    k = r['Kernel_Name']
    name = 'stream_extend' if 'stream_extend<false' in k else 'stream_shade' if 'stream_shade<false' in k else None
    if not name: continue
    agg[name] += float(r['Counter_Value'])
    if (name, r['Dispatch_Id']) not in seen: seen.add((name, r['Dispatch_Id'])); n[name] += 1
print(json.dumps({k: {'launches': n[k], 'WRITE_SIZE_KB_sum': agg[k], 'hbm_write_bytes_per_launch': 1024 * agg[k] / max(1, n[k])} for k in agg}))
PY
rm -rf $R/gpurun_out/pmcW
