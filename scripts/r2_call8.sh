R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2h
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r2h/smoke.txt
grep -q "smoke OK" gpurun_out/r2h/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
for i in 1 2; do ZR_COMMIT_STATS=1 ZR_BVH_PROFILE=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 8 2>&1 | grep "\[zr\]\|bvh_build_upload_s" | sed 's/.*"bvh_build_upload_s": \([0-9.]*\).*/bvh_build_upload_s \1/'; done | tee gpurun_out/r2h/commit_stats.txt
bash scripts/r2_tests.sh
