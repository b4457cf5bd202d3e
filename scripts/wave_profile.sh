set -e
R=$GRAFT_REPO_ROOT
touch $R/raytracer_project_amd/csrc/zr_stream.hip
make -s -C $R/raytracer_project_amd/csrc ZR_KFLAGS=-DZR_WAVE_PROFILE > /dev/null
ZR_STREAM_POOLS=1 python3 $R/scripts/wave_profile.py "$@"
