cd $GRAFT_REPO_ROOT
export ZR_DIST_BACKEND=gloo ZR_BENCH_ONE_DEVICE=1 ZR_STREAM_SLOTS=$((8*1024*1024))
python bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('N=1', d['value'], d['config']['segments_per_step'], d['config']['frame_checksum'])"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 0 --spp 32 2>gpurun_out/n2.err > gpurun_out/n2.out; cat gpurun_out/n2.out | cut -c1-600
tail -3 gpurun_out/n2.err
