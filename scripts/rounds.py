"""Per-round EXTEND durations of the streaming pipeline (development aid, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_project_amd import capi
ctx = capi.Context(0)
for a in sys.argv[1:] or ["cfg3:256", "cfg2:256"]:
    name, spp = a.split(":")
    ds = capi.DemoScene(name); cam = ds.camera.copy(); cam.samples_per_pixel = int(spp)
    sc = capi.Scene(ctx, ds.desc)
    sc.render(cam, ds.env, ds.seed, None); ctx.kernel_times_ms(100000)
    sc.render(cam, ds.env, ds.seed, None); c = ctx.counters(); t = ctx.kernel_times_ms(100000)
    print(name, 'rounds', c.rounds, 'extend total %.1f shade total %.1f' % (c.extend_ms, c.shade_ms))
    print('  extend ms per round:', ' '.join('%.2f' % x for x in t[:12]), '...', ' '.join('%.2f' % x for x in t[len(t)//2-3:len(t)//2+3]), '...', ' '.join('%.2f' % x for x in t[-12:]))
