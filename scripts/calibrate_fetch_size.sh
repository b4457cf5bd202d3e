# FETCH_SIZE calibration on this access pattern (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own
# access pattern"): dependent random reads of 32 / 64 / 128-byte records from a 1 GB array (no reuse), scripts/dev/randread.hip
set -e
R=$GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -w $R/scripts/dev/randread.hip -o /tmp/randread
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/cal
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/cal -- /tmp/randread > $R/gpurun_out/cal_stdout.txt 2>&1
python3 - <<'PY'
import csv, glob, os, re
R = os.environ['GRAFT_REPO_ROOT']
f = sorted(glob.glob(f'{R}/gpurun_out/cal/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == 'FETCH_SIZE']
# dispatch order = the program's loop order: sizes x lanes x record sizes x 2 repetitions
print('dispatches', len(rows))
for r in rows:
    m = re.search(r'chase<(\d+)>', r['Kernel_Name'])
    print(r['Dispatch_Id'], 'record', m.group(1) if m else '?', 'FETCH_SIZE_KB', r['Counter_Value'])
PY
rm -rf $R/gpurun_out/cal
