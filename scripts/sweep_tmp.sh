set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc*
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline > $R/gpurun_out/pmc1.json 2> $R/gpurun_out/pmc1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline > $R/gpurun_out/pmc2.json 2> $R/gpurun_out/pmc2.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline > $R/gpurun_out/pmc3.json 2> $R/gpurun_out/pmc3.err || true
