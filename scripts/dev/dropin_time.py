import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raytracer_project_amd import capi
for name in ("cfg3", "cfg5"):
    t0 = time.perf_counter(); ds = capi.DemoScene(name); t1 = time.perf_counter()
    print(name, "scene build (generate + flatten for desc)", round(t1 - t0, 3), "s")
    for k in range(3):
        t0 = time.perf_counter(); a, ctr = ds.render_dropin(spp=1); t1 = time.perf_counter()
        print(name, "drop-in camera::render at 1 spp:", round((t1 - t0) * 1e3, 1), "ms  (kernel", round(ctr.kernel_ms, 2), "ms)")
