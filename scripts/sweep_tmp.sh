cd $GRAFT_REPO_ROOT
for cfg in "-DST_EXT_WAVES=4 -DST_SHADE_WAVES=2" "-DST_EXT_WAVES=5 -DST_SHADE_WAVES=2" "-DST_EXT_WAVES=6 -DST_SHADE_WAVES=2" "-DST_EXT_WAVES=4 -DST_SHADE_WAVES=3" "-DST_EXT_WAVES=4 -DST_SHADE_WAVES=4"; do
  touch raytracer_project_amd/csrc/zr_stream.hip
  make -s -C raytracer_project_amd/csrc "ZR_KFLAGS=$cfg" 2>&1 | grep -E "error" || true
  echo "== $cfg"
  python scripts/stats.py cfg3:256 2>&1 | head -1
done
