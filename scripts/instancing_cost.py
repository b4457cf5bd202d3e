"""Development aid (GPU): what the two-level BVH costs and saves on inst0 at a frame size that keeps the GPU busy — the same scene
flattened with groups (each mesh stored once, placements walked through the mapped ray) and with ZR_GROUPS=0 (every placement a
baked world-space copy in the one tree)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_project_amd import capi
for groups in ("1", "0"):
    os.environ["ZR_GROUPS"] = groups
    ds = capi.DemoScene("inst0")
    cam = ds.camera.copy(); cam.image_width, cam.image_height, cam.samples_per_pixel = 1920, 1080, 64
    c = capi.Context(0)
    t0 = time.perf_counter(); sc = capi.Scene(c, ds.desc); t_commit = time.perf_counter() - t0
    sc.render(cam, ds.env, ds.seed, None)
    t0 = time.perf_counter(); sc.render(cam, ds.env, ds.seed, None, count=False); dt = time.perf_counter() - t0
    sc.render(cam, ds.env, ds.seed, None, count=True); k = c.counters()
    print(f"ZR_GROUPS={groups}: world entries {ds.desc.n_objects}, groups {ds.desc.n_groups}, device bytes {sc.stats()['device_bytes']}, commit {t_commit*1e3:.1f} ms, "
          f"frame {dt*1e3:.1f} ms, {k.segments/dt*1e-6:.0f} Msegments/s, extend {k.extend_ms:.1f} ms shade {k.shade_ms:.1f} ms over {k.rounds} rounds")
    sc.close(); c.close()
