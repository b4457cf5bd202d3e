R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2i
ZR_COMMIT_STATS=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 8 2>&1 | grep "\[zr\] commit\|bvh_build_upload_s" | sed 's/.*"bvh_build_upload_s": \([0-9.]*\).*/bvh_build_upload_s \1/' | tee gpurun_out/r2i/commit_stats.txt
BENCH_ARGS="--workload cfg3" timeout -k 10 700 bash scripts/ab_flags.sh "-DST_SHADE_WAVES=5" "-DST_SHADE_WAVES=6" "-DST_SHADE_WAVES=3" "-DST_EXT_WAVES_LEAN=8" "-DST_EXT_WAVES_LEAN=5" 2>&1 | tee gpurun_out/r2i/ab_waves.txt
