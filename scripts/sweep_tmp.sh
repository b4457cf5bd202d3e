cd $GRAFT_REPO_ROOT
for w in 5 6 7 8; do
  touch raytracer_project_amd/csrc/zr_stream.hip
  make -s -C raytracer_project_amd/csrc "ZR_KFLAGS=-DST_EXT_WAVES_LEAN=$w" 2>&1 | grep -E "error" || true
  echo "== LEAN waves $w"
  python scripts/stats.py cfg3:256 cfg2:128 2>&1 | grep Mseg
done
