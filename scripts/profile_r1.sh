# rocprofv3 evidence for profiles/ (run on the GPU box through gpurun from the repo root)
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r1_v2}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err
# HBM traffic of the same command: FETCH_SIZE in its own pass (3 of the 4 TCC slots), counters only with --kernel-trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_bench.json 2> $R/gpurun_out/pmc_$TAG.err
ls $R/gpurun_out/prof_$TAG/* $R/gpurun_out/pmc_$TAG/*
