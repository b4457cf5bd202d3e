R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c17; mkdir -p $O
cd $R
S=/tmp/zr_ab17
mkdir -p $S/raytracer_project_amd && cp -r $R/include $S/ && cp -r $R/scenes $S/ && cp -r $R/raytracer_project_amd/csrc $S/raytracer_project_amd/
touch $S/raytracer_project_amd/csrc/*.hip
make -s -j8 -C $S/raytracer_project_amd/csrc ZR_DEFS="-DST_NODE_MINMAX" > $S/build.log 2>&1 || { echo BUILD FAILED; tail -5 $S/build.log; exit 1; }
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "extend ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"], "launches", r["launches_timed"], "checksum", d["config"]["frame_checksum"])'; }
for k in 1 2; do
echo "new: $(run)" | tee -a $O/ab.txt
echo "minmax: $(ZR_LIB=$S/raytracer_project_amd/csrc/libzr_hip.so run)" | tee -a $O/ab.txt
done
cd /tmp; export TMPDIR=/tmp
for v in new minmax; do
  if [ $v = minmax ]; then export ZR_LIB=$S/raytracer_project_amd/csrc/libzr_hip.so; fi
  rm -rf $O/pmc_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$v -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/pmc_$v.err || echo "pmc pass failed"
  python3 - $O/pmc_$v $v <<'PY' | tee -a $O/ab.txt
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'].split('(')[0][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_INSTS_VALU': n[k] += 1
for k in acc:
    if 'stream_extend' in k or 'stream_shade' in k:
        print(sys.argv[2], k, n[k], {c: '%.4g' % (v / max(n[k], 1)) for c, v in acc[k].items()})
PY
  rm -rf $O/pmc_$v
done
