R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c36; mkdir -p $O
cd $R
echo "== cfg3 whole frame" | tee $O/ab.txt
BENCH_STEPS=4 bash scripts/ab_flags.sh "-DST_TAIL_CHUNKS=0" "-DST_TAIL_CHUNKS=32" "-DST_TAIL_CHUNKS=192" 2>&1 | tee -a $O/ab.txt
echo "== cfg3 1/8 share" | tee -a $O/ab.txt
ZR_BENCH_SHARD_OF=8 BENCH_STEPS=6 bash scripts/ab_flags.sh "-DST_TAIL_CHUNKS=0" "-DST_TAIL_CHUNKS=32" "-DST_TAIL_CHUNKS=192" 2>&1 | tee -a $O/ab.txt
