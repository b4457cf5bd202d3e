cd $GRAFT_REPO_ROOT
for cfg in "-DST_FETCH_MIN=16" "-DST_FETCH_MIN=8" "-DST_FETCH_MIN=32" "-DST_FETCH_MIN=48" "-DST_CHUNK=128" "-DST_CHUNK=512" "-DST_LDS_STACK=8" "-DST_LDS_STACK=16"; do
  touch raytracer_project_amd/csrc/zr_stream.hip
  make -s -C raytracer_project_amd/csrc "ZR_KFLAGS=$cfg" 2>&1 | grep -E "error" || true
  echo "== $cfg"; python scripts/stats.py cfg3:256 | grep Mseg | cut -c1-110
done
