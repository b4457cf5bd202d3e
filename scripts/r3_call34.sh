R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c34; mkdir -p $O
cd $R
run() { python3 $R/bench.py --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("steps", d["steps"], "ms_per_step", d["ms_per_step"], "extend ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"])'; }
for k in 1 2; do
run 5 2 | tee -a $O/ab.txt
run 20 5 | tee -a $O/ab.txt
run 60 5 | tee -a $O/ab.txt
done
rocm-smi --showclocks --showpower 2>/dev/null | head -30 | tee -a $O/ab.txt
