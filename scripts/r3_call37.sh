R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c37; mkdir -p $O
cd $R
BENCH_STEPS=3 bash scripts/ab_kflags.sh "-O2" "-fno-unroll-loops" "-mllvm -amdgpu-schedule-relaxed-occupancy=true" "-mllvm -enable-post-misched=false" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=true" "-mllvm -amdgpu-use-amdgpu-trackers=true" "-mllvm -greedy-reverse-local-assignment=true" "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=true" 2>&1 | tee $O/ab.txt
