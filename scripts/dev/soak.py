"""Development aid: restart soak — the reference rebuilds geometry + BVH and starts a fresh render thread on every restart (main.cpp:1492-1531).
N restarts of the drop-in's camera::render on cfg3 at 2 spp (flatten + commit + render each time, each on a thread of its own), device
memory watched: it must not grow, and every frame must equal the first."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from raytracer_project_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ds = capi.DemoScene("cfg3")
first = None
t0 = time.perf_counter()
for k in range(n):
    a, ctr = ds.render_dropin(spp=2)
    if first is None:
        first = a.copy()
    elif not np.array_equal(a, first):
        print("frame", k, "differs from the first"); sys.exit(1)
    if k % 10 == 0 or k == n - 1:
        free, total = torch.cuda.mem_get_info(0)
        print(f"restart {k:3d}: device memory in use {(total - free) / 2**30:7.2f} GiB, {(time.perf_counter() - t0) / (k + 1) * 1e3:7.1f} ms per restart so far", flush=True)
print("soak OK")
