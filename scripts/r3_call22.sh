R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3eb; mkdir -p $O
cd $R
for n in 2 4; do
  ZR_BENCH_ONE_DEVICE=1 ZR_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline > $O/r3_rehearsal_${n}ranks_one_gpu_gloo.json 2> $O/rehearsal_${n}.err; echo "rehearsal $n ranks exit $?"
  head -c 200 $O/r3_rehearsal_${n}ranks_one_gpu_gloo.json; echo
done
