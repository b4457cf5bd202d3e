"""Development aid (GPU): per-round EXTEND and SHADE durations of one rank's 1/N share of cfg3 (one pool, so that a launch's duration is
that kernel alone): where does a short frame lose time against 1/N of the whole frame?"""
import os, sys
os.environ["ZR_STREAM_POOLS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_project_amd import capi, multi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ds = capi.DemoScene(os.environ.get("ZR_ROUNDS_SCENE", "cfg3")); cam = ds.camera
region = multi.tile_region(capi, 0, n) if n > 1 else None
out = {}
for kind, tag in ((1, "extend"), (2, "shade")):
    os.environ["ZR_TIMELOG_KIND"] = str(kind)
    ctx = capi.Context(0)
    sc = capi.Scene(ctx, ds.desc)
    sc.render(cam, ds.env, ds.seed, region); ctx.kernel_times_ms(100000)
    sc.render(cam, ds.env, ds.seed, region); c = ctx.counters(); out[tag] = ctx.kernel_times_ms(100000)
    print(tag, 'rounds', c.rounds, 'extend total %.1f shade total %.1f' % (c.extend_ms, c.shade_ms))
    sc.close(); ctx.close()
for tag, t in out.items():
    print(tag, 'ms per round:', ' '.join('%.2f' % x for x in t))
