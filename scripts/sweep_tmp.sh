cd $GRAFT_REPO_ROOT
for cfg in "ZR_BVH_MAX_LEAF=4" "ZR_BVH_MAX_LEAF=2" "ZR_BVH_MAX_LEAF=8" "ZR_BVH_MAX_LEAF=8 ZR_BVH_COST_TRI=0.7" "ZR_BVH_COST_TRI=3" "ZR_BVH_COST_TRAVERSE=2" "ZR_BVH_COST_TRAVERSE=0.5" "ZR_BVH_MAX_LEAF=16 ZR_BVH_COST_TRI=0.5"; do
echo "== $cfg"; env $cfg python scripts/stats.py cfg3:256 | grep Mseg | cut -c1-140
done
