# The evidence set of a round (run on the GPU box through gpurun from the repo root):
#   bash scripts/profile_round.sh <tag> [workload]        e.g.  bash scripts/profile_round.sh r2 cfg3
# 1. rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1`  -> gpurun_out/<tag>_<workload>_kernel_stats.csv + the bench line
# 2. FETCH_SIZE and WRITE_SIZE of the same command, each in its OWN --pmc pass (they do not fit one pass on gfx950; counters only with
#    --kernel-trace)                                                      -> gpurun_out/<tag>_<workload>_traffic.json, keyed by the kernel
#    source hash (bench.py --kernel-src-sha) so that bench.py only quotes it for the sources it was taken on
# Copy the summaries into profiles/ by hand.
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}; WL=${2:-cfg3}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_${TAG}_$WL $OUT/pmcF_${TAG}_$WL $OUT/pmcW_${TAG}_$WL $OUT/pmcL_${TAG}_$WL $OUT/pmcD_${TAG}_$WL
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$WL -- python3 $R/bench.py --steps 2 --warmup 1 --workload $WL --no-cpu-baseline > $OUT/${TAG}_${WL}_bench_profiled.json 2> $OUT/prof_${TAG}_$WL.err || echo "stats pass failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcF_${TAG}_$WL -- python3 $R/bench.py --steps 1 --warmup 0 --workload $WL --no-cpu-baseline > /dev/null 2> $OUT/pmcF_${TAG}_$WL.err || echo "FETCH_SIZE pass failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmcW_${TAG}_$WL -- python3 $R/bench.py --steps 1 --warmup 0 --workload $WL --no-cpu-baseline > /dev/null 2> $OUT/pmcW_${TAG}_$WL.err || echo "WRITE_SIZE pass failed"
# 3. L2 (TCC) hits / misses and the FP64 vector instructions of the same command, one --pmc pass each -> the same JSON (bench.py: roofline.l2_hit_rate,
#    roofline.fp64_valu_frac)
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmcL_${TAG}_$WL -- python3 $R/bench.py --steps 1 --warmup 0 --workload $WL --no-cpu-baseline > /dev/null 2> $OUT/pmcL_${TAG}_$WL.err || echo "TCC pass failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcD_${TAG}_$WL -- python3 $R/bench.py --steps 1 --warmup 0 --workload $WL --no-cpu-baseline > /dev/null 2> $OUT/pmcD_${TAG}_$WL.err || echo "FP64 pass failed"
SHA=$(python3 $R/bench.py --kernel-src-sha)
python3 - "$OUT" "$TAG" "$WL" "$SHA" <<'PY'
import csv, glob, json, os, sys, collections
out, tag, wl, sha = sys.argv[1:5]
f = sorted(glob.glob(f'{out}/prof_{tag}_{wl}/*/*kernel_stats.csv'), key=os.path.getmtime)
if f: open(f'{out}/{tag}_{wl}_kernel_stats.csv', 'w').write(open(f[-1]).read())
def sums(d, counter):
    fs = sorted(glob.glob(f'{out}/{d}_{tag}_{wl}/*/*counter_collection.csv'), key=os.path.getmtime)
    agg = collections.defaultdict(float); n = collections.Counter(); seen = set()
    if not fs: return agg, n
    for r in csv.DictReader(open(fs[-1])):
        if r['Counter_Name'] != counter: continue
        k = r['Kernel_Name']
        name = 'stream_extend' if 'stream_extend<false' in k else 'stream_shade' if 'stream_shade<false' in k else 'fused_render' if ('fused_render<' in k and 'false' in k) else None
        if not name: continue
        agg[name] += float(r['Counter_Value'])
        if (name, r['Dispatch_Id']) not in seen: seen.add((name, r['Dispatch_Id'])); n[name] += 1
    return agg, n
fa, fn = sums('pmcF', 'FETCH_SIZE'); wa, wn = sums('pmcW', 'WRITE_SIZE')
th, tn = sums('pmcL', 'TCC_HIT_sum'); tm, _ = sums('pmcL', 'TCC_MISS_sum'); tr, _ = sums('pmcL', 'TCC_REQ_sum')
d_fma, dn = sums('pmcD', 'SQ_INSTS_VALU_FMA_F64'); d_mul, _ = sums('pmcD', 'SQ_INSTS_VALU_MUL_F64'); d_add, _ = sums('pmcD', 'SQ_INSTS_VALU_ADD_F64'); d_all, _ = sums('pmcD', 'SQ_INSTS_VALU'); d_gui, _ = sums('pmcD', 'GRBM_GUI_ACTIVE')
res = {'workload': wl, 'kernel_src_sha': sha,
       'command': f'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one pass each) --output-format csv -- python3 bench.py --steps 1 --warmup 0 --workload {wl} --no-cpu-baseline',
       'note': 'FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section: exact for coalesced rows, an upper bound for the 64-byte node requests: profiles/README.md); Infinity-Cache hits are counted, not excluded; WRITE_SIZE x 1024 as is',
       'kernels': {}}
for k in sorted(set(fa) | set(wa) | set(th) | set(d_fma)):
    res['kernels'][k] = {'launches': fn.get(k, wn.get(k, 0)), 'FETCH_SIZE_KB_sum': fa.get(k, 0.0), 'hbm_read_bytes_per_launch': 2 * 1024 * fa.get(k, 0.0) / max(1, fn.get(k, 0)),
                         'WRITE_SIZE_KB_sum': wa.get(k, 0.0), 'hbm_write_bytes_per_launch': 1024 * wa.get(k, 0.0) / max(1, wn.get(k, 0))}
    if th.get(k, 0) + tm.get(k, 0) > 0:
        res['kernels'][k].update({'TCC_HIT_sum': th[k], 'TCC_MISS_sum': tm.get(k, 0.0), 'TCC_REQ_sum': tr.get(k, 0.0), 'l2_hit_rate': round(th[k] / (th[k] + tm.get(k, 0.0)), 4),
                                  'l2_requests_per_launch': tr.get(k, 0.0) / max(1, tn.get(k, 0))})
    if dn.get(k, 0):
        # wave-level instruction counts: an FP64 FMA is 2 flops on each of 64 lanes (inactive lanes are counted too: an upper bound on useful work)
        res['kernels'][k].update({'SQ_INSTS_VALU_FMA_F64': d_fma.get(k, 0.0), 'SQ_INSTS_VALU_MUL_F64': d_mul.get(k, 0.0), 'SQ_INSTS_VALU_ADD_F64': d_add.get(k, 0.0),
                                  'SQ_INSTS_VALU': d_all.get(k, 0.0), 'GRBM_GUI_ACTIVE': d_gui.get(k, 0.0), 'valu_insts_per_launch': d_all.get(k, 0.0) / dn[k],
                                  # share of the chip's vector issue slots the kernel used: a wave64 VALU instruction occupies its SIMD for 4 cycles;
                                  # GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1024 SIMDs on the chip
                                  'valu_issue_frac': round(4.0 * d_all.get(k, 0.0) / (d_gui[k] / 8.0 * 1024.0), 4) if d_gui.get(k, 0) else None,
                                  'fp64_flops_per_launch': 64.0 * (2 * d_fma.get(k, 0.0) + d_mul.get(k, 0.0) + d_add.get(k, 0.0)) / dn[k]})
json.dump(res, open(f'{out}/{tag}_{wl}_traffic.json', 'w'), indent=1)
print(json.dumps(res['kernels']))
PY
rm -rf $OUT/prof_${TAG}_$WL $OUT/pmcF_${TAG}_$WL $OUT/pmcW_${TAG}_$WL $OUT/pmcL_${TAG}_$WL $OUT/pmcD_${TAG}_$WL
