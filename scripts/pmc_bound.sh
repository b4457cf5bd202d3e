# What bounds the pipeline kernels?  Three --pmc passes (counters only, with --kernel-trace) over one cfg3 frame at reduced spp:
# SQ issue / wait accounting, texture-addresser (TA) and data-return (TD) busy cycles, L1 (TCP) stalls and request latency.
#   bash scripts/pmc_bound.sh <out_dir> [spp] [workload]
R=$GRAFT_REPO_ROOT; OUT=${1:-$R/gpurun_out/pmc_bound}; SPP=${2:-64}; WL=${3:-cfg3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P() { tag=$1; shift; rm -rf $OUT/raw_$tag; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/raw_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --spp $SPP --workload $WL --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/err_$tag.txt || echo "pass $tag failed: $(tail -2 $OUT/err_$tag.txt)"; }
P sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM
P sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
# (round 2 asked for seven TA / TD and seven TCP counters in one pass each: rocprofv3 aborted with "Request exceeds the capabilities of the
#  hardware to collect"; round 3's first try with four TA counters aborted the same way: the TA block takes two at a time, TD and TCP four)
P ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
P ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
P td TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
P tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
P tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
P tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
python3 - "$OUT" <<'PY'
import csv, glob, collections, os, sys, json
out = sys.argv[1]
agg = collections.defaultdict(float); n = collections.Counter()
for tag in ("sq", "sq2", "ta", "ta2", "td", "tcp", "tcp2", "tcc"):
    fs = sorted(glob.glob(f"{out}/raw_{tag}/*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs: continue
    seen = set()
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        name = "stream_extend" if "stream_extend<false" in k else "stream_shade" if "stream_shade<false" in k else "fused_render" if "fused_render<" in k and "false" in k else None
        if not name: continue
        agg[(name, r["Counter_Name"])] += float(r["Counter_Value"])
        if (name, r["Counter_Name"], r["Dispatch_Id"]) not in seen: seen.add((name, r["Counter_Name"], r["Dispatch_Id"])); n[(name, r["Counter_Name"])] += 1
res = {}
for (k, c), v in sorted(agg.items()):
    res.setdefault(k, {})[c] = {"sum": v, "launches": n[(k, c)]}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
for k in res:
    print("==", k)
    for c, d in res[k].items(): print("  %-44s %.5g  (%d launches)" % (c, d["sum"], d["launches"]))
PY
rm -rf $OUT/raw_*
