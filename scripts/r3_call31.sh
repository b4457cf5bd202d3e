R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c31; mkdir -p $O
cd $R
for cfg in "ZR_STREAM_POOLS=2" "ZR_STREAM_POOLS=3" "ZR_STREAM_AFFINE=1" "ZR_STREAM_SLOTS=1048576" "ZR_FUSED=0" "ZR_BVH_DEVICE_MIN=0" "ZR_STREAM_DRAIN_POOL=0" "ZR_BVH_TOP=0 ZR_BVH_DEVICE_MIN=0" "ZR_BVH_PLOC_RADIUS=4 ZR_BVH_DEVICE_MIN=0"; do
  n=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest_$n.txt 2>&1; rc=$?
  echo "$cfg: exit $rc: $(tail -1 $O/pytest_$n.txt)" | tee -a $O/summary.txt
  if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_$n.txt | head -8 | tee -a $O/summary.txt; fi
done
