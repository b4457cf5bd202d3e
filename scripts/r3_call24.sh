R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c24; mkdir -p $O
cd $R
timeout -k 10 500 python3 scripts/dev/soak.py 60 2>&1 | grep -v "^\[zenith\]\|Model:" | tee $O/soak.txt
timeout -k 10 300 python3 bench.py --steps 150 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print("150 frames:", d["ms_per_step"], d["value"], d["config"]["frame_checksum"])' | tee -a $O/soak.txt
