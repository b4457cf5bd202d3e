# round 3, call 1: GPU suite on the affine hand-out, then A/B of the hand-out on cfg3 (whole frame and a 1/8 share), L2 hit rates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c1; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d scripts/dev/h2d.hip && /tmp/h2d 160 > $O/h2d.txt 2>&1; cat $O/h2d.txt
B() { env "$@" python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "extend ms/launch", r["kernel_ms"], "launches/step", r["launches_timed"]//d["steps"], "extend per step", r["kernel_ms_per_step"], "checksum", d["config"]["frame_checksum"], "commit_s", d["config"]["bvh_build_upload_s"])'; }
{
echo "== cfg3 whole frame"
for kv in ZR_STREAM_AFFINE=0 ZR_STREAM_AFFINE=1 "ZR_STREAM_AFFINE=1 ZR_STREAM_CHUNK_PX=256" "ZR_STREAM_AFFINE=1 ZR_STREAM_CHUNK_PX=4096" "ZR_STREAM_AFFINE=1 ZR_STREAM_BOTTOM_UP=0"; do echo "$kv: $(B $kv)"; done
echo "== cfg3 1/8 share"
for kv in ZR_STREAM_AFFINE=0 ZR_STREAM_AFFINE=1 "ZR_STREAM_AFFINE=1 ZR_STREAM_CHUNK_PX=256"; do echo "$kv: $(B ZR_BENCH_SHARD_OF=8 $kv)"; done
echo "== cfg5 / cfg2 / demo"
for w in cfg5 cfg2 demo; do for kv in ZR_STREAM_AFFINE=0 ZR_STREAM_AFFINE=1; do echo "$w $kv: $(env $kv python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "value", d["value"], "checksum", d["config"]["frame_checksum"])')"; done; done
} > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for aff in 0 1; do
  rm -rf $O/pmcL
  ZR_STREAM_AFFINE=$aff timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $O/pmcL -- python3 $R/bench.py --steps 1 --warmup 0 --spp 64 --no-cpu-baseline > $O/pmcL_$aff.json 2> $O/pmcL_$aff.err
  echo "== affine=$aff (cfg3, 64 spp)" >> $O/l2.txt
  python3 - $O >> $O/l2.txt <<'PY'
import csv,glob,collections,os,sys
O=sys.argv[1]
f=sorted(glob.glob(f'{O}/pmcL/*/*counter_collection.csv'),key=os.path.getmtime)[-1]
agg=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    name='extend' if 'stream_extend' in k else 'shade' if 'stream_shade' in k else None
    if not name or '<true' in k: continue
    agg[(name,r['Counter_Name'])]+=float(r['Counter_Value'])
for k in sorted(agg): print(k,'%.4g'%agg[k])
for n in ('extend','shade'):
    h,m=agg.get((n,'TCC_HIT_sum'),0),agg.get((n,'TCC_MISS_sum'),0)
    if h+m: print(n,'L2 hit rate %.3f'%(h/(h+m)))
PY
done
rm -rf $O/pmcL
cat $O/l2.txt
