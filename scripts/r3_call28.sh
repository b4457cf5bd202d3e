R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c28; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/trace
ZR_BENCH_SHARD_OF=8 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/err.txt
python3 - $O/trace <<'PY' | tee $O/timeline.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void zr::','')[:28]) for r in rows), key=lambda x: x[0])
# last frame: find the last stream_init
idx = [i for i, e in enumerate(ev) if 'stream_init' in e[2]]
start = idx[-2] if len(idx) >= 2 else idx[-1]   # first init of the last frame (2 pools -> 2 inits)
fr = ev[start:]
t0 = fr[0][0]
busy_end = t0; idle = 0; 
print("kernels in last frame:", len(fr), "span ms", (max(e[1] for e in fr) - t0) / 1e6)
# union busy time
iv = sorted((e[0], e[1]) for e in fr)
cur_s, cur_e = iv[0]; busy = 0; gaps = []
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; gaps.append((cur_e - t0, s - cur_e)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("GPU busy (union) ms", busy / 1e6, "idle gaps ms", sum(g for _, g in gaps) / 1e6, "n gaps", len(gaps))
print("largest gaps (at ms, len us):", [(round(a / 1e6, 2), round(g / 1e3, 1)) for a, g in sorted(gaps, key=lambda x: -x[1])[:12]])
by = {}
for s, e, n in fr: by.setdefault(n, [0, 0]); by[n][0] += 1; by[n][1] += e - s
for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1]): print(f"{n:30s} {c:4d} launches {t / 1e6:8.2f} ms total {t / c / 1e3:8.1f} us avg")
for s, e, n in fr[:60]: print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  {n}")
PY
rm -rf $O/trace
