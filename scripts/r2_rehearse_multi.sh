# rehearsal of bench.py's N > 1 path on a one-GPU box: 2 and 4 ranks share device 0, gloo carries the packed-tile all-gather
# (the real run uses RCCL, one GPU per rank); prints the JSON line of each
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2v
for n in 2 4; do
  ZR_BENCH_ONE_DEVICE=1 ZR_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2v/multi$n.err | tail -1 | tee gpurun_out/r2v/multi$n.json | cut -c1-400
done
