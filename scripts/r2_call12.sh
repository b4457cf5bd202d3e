R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2l
for w in cfg3 demo; do BENCH_ARGS="--workload $w" timeout -k 10 500 bash scripts/ab_flags.sh "-DZR_EXT_TOUCH=1" "-DZR_EXT_TOUCH=2" 2>&1 | sed "s/^/$w /"; done | tee gpurun_out/r2l/ext_touch.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2l/prof -- python3 $R/bench.py --steps 2 --warmup 1 --workload cfg3 --no-cpu-baseline > $R/gpurun_out/r2l/bench_profiled.json 2> $R/gpurun_out/r2l/prof.err
cp $(ls $R/gpurun_out/r2l/prof/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r2l/kernel_stats.csv
head -8 $R/gpurun_out/r2l/kernel_stats.csv | cut -c1-160
