# development aid: how busy the persistent EXTEND waves are.  Builds a -DZR_WAVE_PROFILE library into a scratch copy of csrc/
# and points capi at it through ZR_LIB: the in-tree library is never replaced by the instrumented build.
set -e
R=$GRAFT_REPO_ROOT
S=${TMPDIR:-/tmp}/zr_wp_$$
mkdir -p $S/raytracer_project_amd && cp -r $R/include $S/ && cp -r $R/scenes $S/ && cp -r $R/raytracer_project_amd/csrc $S/raytracer_project_amd/
touch $S/raytracer_project_amd/csrc/zr_stream.hip
make -s -j8 -C $S/raytracer_project_amd/csrc ZR_KFLAGS=-DZR_WAVE_PROFILE > /dev/null
ZR_LANE_HISTOGRAM=1 ZR_LIB=$S/raytracer_project_amd/csrc/libzr_hip.so ZR_STREAM_POOLS=1 python3 $R/scripts/wave_profile.py "$@"
rm -rf $S
