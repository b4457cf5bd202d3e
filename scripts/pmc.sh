# PMC snapshot of the pipeline kernels (run on the GPU box via gpurun): bash scripts/pmc.sh <spp>
set -e
R=$GRAFT_REPO_ROOT; SPP=${1:-64}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcA $R/gpurun_out/pmcB
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline > $R/gpurun_out/pmcA.json 2> $R/gpurun_out/pmcA.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcB -- python3 $R/bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline > $R/gpurun_out/pmcB.json 2> $R/gpurun_out/pmcB.err
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
for d in ['pmcA','pmcB']:
    f=sorted(glob.glob(f'{R}/gpurun_out/{d}/*/*counter_collection.csv'),key=os.path.getmtime)[-1]
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        name='extend' if 'stream_extend' in k else 'shade' if 'stream_shade' in k else None
        if not name or '<true' in k: continue
        agg[(name,r['Counter_Name'])]+=float(r['Counter_Value']); n[(name,r['Counter_Name'])]+=1
    for k in sorted(agg): print(k,'%.4g'%agg[k],n[k])
PY
