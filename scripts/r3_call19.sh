R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c19; mkdir -p $O
cd $R
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "extend ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"], "launches", r["launches_timed"], "checksum", d["config"]["frame_checksum"])'; }
echo "default: $(run)" | tee -a $O/ab.txt
echo "affine: $(ZR_STREAM_AFFINE=1 run)" | tee -a $O/ab.txt
echo "pools2: $(ZR_STREAM_POOLS=2 run)" | tee -a $O/ab.txt
echo "affine+pools2: $(ZR_STREAM_AFFINE=1 ZR_STREAM_POOLS=2 run)" | tee -a $O/ab.txt
echo "pools4: $(ZR_STREAM_POOLS=4 run)" | tee -a $O/ab.txt
echo "default: $(run)" | tee -a $O/ab.txt
