# development aid: kernel timeline of a 1/8-shard frame (rocprofv3 --kernel-trace), env passes through
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl8
ZR_BENCH_SHARD_OF=${SHARD:-8} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl8 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/tl8.json 2> $R/gpurun_out/tl8.err
f=$(ls -t $R/gpurun_out/tl8/*/*kernel_trace.csv | head -1)
python3 $R/scripts/timeline.py $f 2
rm -rf $R/gpurun_out/tl8
