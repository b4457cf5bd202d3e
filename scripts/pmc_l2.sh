# development aid: L2 hit/miss of the pipeline kernels, full frame vs 1/8 shard (64 spp)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in full shard; do
  rm -rf $R/gpurun_out/pmcL
  if [ $mode = shard ]; then export ZR_BENCH_SHARD_OF=8; SPP=512; else unset ZR_BENCH_SHARD_OF; SPP=64; fi
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmcL -- python3 $R/bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline > $R/gpurun_out/pmcL.json 2> $R/gpurun_out/pmcL.err
  echo "== $mode"
  python3 - <<'PY'
import csv,glob,collections,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
f=sorted(glob.glob(f'{R}/gpurun_out/pmcL/*/*counter_collection.csv'),key=os.path.getmtime)[-1]
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
rm -rf $R/gpurun_out/pmcL
