# round-2 evidence, part a: smoke, commit phases, the whole GPU test suite
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2g
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2g/smoke.txt
grep -q "smoke OK" gpurun_out/r2g/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
ZR_COMMIT_STATS=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 8 2>&1 >/dev/null | grep "\[zr\]" | tee gpurun_out/r2_commit_stats.txt
bash scripts/r2_tests.sh
