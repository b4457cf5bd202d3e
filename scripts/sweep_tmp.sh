cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for ov in 0 1; do for b in 4096 6144; do
echo "== overlap=$ov blocks=$b"
ZR_STREAM_OVERLAP=$ov ZR_ST_BLOCKS=$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
done; done
