R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2n
ZR_COMMIT_STATS=1 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 4 2>&1 >/dev/null | grep "workgroups" | tee gpurun_out/r2n/occupancy.txt
BENCH_ARGS="--workload cfg3" timeout -k 10 900 bash scripts/ab_flags.sh "-DST_EXT_GROUP=1 -DZR_EXT_NO_TOP -DST_LDS_STACK=12" "-DST_EXT_GROUP=8 -DZR_EXT_NO_TOP" "-DST_EXT_GROUP=4 -DZR_EXT_NO_TOP -DST_LDS_STACK=12" "-DST_EXT_GROUP=2 -DZR_EXT_NO_TOP -DST_LDS_STACK=12" 2>&1 | tee gpurun_out/r2n/group_bisect.txt
