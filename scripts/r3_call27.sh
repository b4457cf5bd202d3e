R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c27; mkdir -p $O
cd $R
S=/tmp/zr_ab27
mkdir -p $S/raytracer_project_amd && cp -r $R/include $S/ && cp -r $R/scenes $S/ && cp -r $R/raytracer_project_amd/csrc $S/raytracer_project_amd/
cp scripts/dev/zr_stream_head.tmp $S/raytracer_project_amd/csrc/zr_stream.hip
touch $S/raytracer_project_amd/csrc/*.hip
make -s -j8 -C $S/raytracer_project_amd/csrc > $S/build.log 2>&1 || { echo BUILD FAILED; tail -5 $S/build.log; exit 1; }
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "extend ms/launch", r["kernel_ms"], "per step", r["kernel_ms_per_step"], "launches", r["launches_timed"], "checksum", d["config"]["frame_checksum"])'; }
for w in cfg3 demo cfg2 cfg3; do
export BENCH_ARGS="--workload $w"
echo "$w new: $(run)" | tee -a $O/ab.txt
echo "$w head: $(ZR_LIB=$S/raytracer_project_amd/csrc/libzr_hip.so run)" | tee -a $O/ab.txt
done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
