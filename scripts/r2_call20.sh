R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2r
BENCH_ARGS="--workload cfg3" timeout -k 10 900 bash scripts/ab_env.sh "ZR_BVH_COST_TRI=1.0" "ZR_BVH_COST_TRI=2.0" "ZR_BVH_COST_TRI=3.0" "ZR_BVH_MAX_LEAF=2" "ZR_BVH_MAX_LEAF=8 ZR_BVH_COST_TRI=1.0" "ZR_BVH_COST_TRAVERSE=0.5" 2>&1 | tee gpurun_out/r2r/sah_sweep.txt
