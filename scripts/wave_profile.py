"""Development aid (GPU, needs a -DZR_WAVE_PROFILE build): how busy are the persistent EXTEND waves?"""
import os, sys, time
os.environ["ZR_RAW_COUNTERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_project_amd import capi
import numpy as np
ctx = capi.Context(0)
ds = capi.DemoScene("cfg3"); cam = ds.camera
sc = capi.Scene(ctx, ds.desc)
out = np.zeros((cam.image_height, cam.image_width, 3))
def run(tag, reg):
    sc.render(cam, ds.env, ds.seed, reg, out=out)
    sc.render(cam, ds.env, ds.seed, reg, out=out)
    c = ctx.counters()
    life_ms = c.shade_execs * 1e-5   # gctr[13]: sum of wave lifetimes in 10 ns ticks
    waves = c.shade_lanes            # gctr[14]: waves launched (all rounds)
    ne, nl, le, ll, fe, fl = c.segments, c.nodes_tested, c.spheres_tested, c.triangles_tested, c.cubes_tested, c.media_tested
    blocks = waves / max(1, c.rounds)
    print('%-22s rounds %d extend %.1f ms  waves/round %.0f  mean wave life / kernel time = %.3f' % (tag, c.rounds, c.extend_ms, blocks, life_ms / max(1e-9, blocks * c.extend_ms)))
    tot = ne + le + fe
    print('    iterations/wave/round %.0f: NODE %.1f%% (%.1f lanes)  LEAF %.1f%% (%.1f lanes)  FETCH %.1f%% (%.1f idle lanes)  us/iteration %.2f' % (
        tot / waves, 100 * ne / tot, nl / max(1, ne), 100 * le / tot, ll / max(1, le), 100 * fe / tot, fl / max(1, fe), life_ms * 1e3 / tot), flush=True)
R = capi.Region
run('full', None)
if "--full-only" in sys.argv:
    sys.exit(0)
run('shard 1/8', R(0, 0, 0, 0, 32, 8, 0, 0))
run('band sky y0=0', R(0, 0, 1920, 135, 32, 0, 0, 0))
run('band knot y0=540', R(0, 540, 1920, 135, 32, 0, 0, 0))
