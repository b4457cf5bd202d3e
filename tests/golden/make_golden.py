#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the GENUINE reference arithmetic.

Runs only in the build container: it executes oracle/_ref/zenith_ref, which oracle/Makefile compiles from
the headers under /root/reference (never copied into this repo) plus oracle/ref_harness.cpp.  The fixtures
are data only — inputs (scene name, region, seed, rays) and expected outputs (radiance, hit records,
segment / RNG-draw counts).  Commit the .npz files together with this script.

    python tests/golden/make_golden.py            # everything (cfg3_full takes ~1 min: 1M-triangle BVH)
    python tests/golden/make_golden.py cfg1 mix0  # selected fixtures
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "zenith_ref")

# name -> (scene, x0, y0, w, h, spp (0 = the config's own), per_sample, extra scene args)
TILES = {
    "cfg1_full": ("cfg1", 0, 0, 400, 225, 0, False, []),             # the whole plumbing config
    "cfg1_tile": ("cfg1", 180, 100, 16, 16, 0, True, []),
    "cfg2_tile": ("cfg2", 600, 300, 16, 16, 0, False, []),            # full 256 spp
    "cfg2_tile_b": ("cfg2", 300, 420, 12, 12, 0, False, []),
    "cfg3_small": ("cfg3", 900, 500, 24, 24, 64, False, [200, 20, 256, 128]),   # 8 000 triangles, 256x128 HDRI
    "cfg3_full": ("cfg3", 944, 528, 16, 16, 0, False, []),            # 1 000 000 triangles, 4096x2048 HDRI, 512 spp
    "cfg5_tile": ("cfg5", 250, 300, 8, 8, 0, False, []),              # 1024 spp, depth 50
    "cfg5_tile_b": ("cfg5", 150, 450, 8, 8, 256, True, []),            # glass sphere region
    "mix0_full": ("mix0", 0, 0, 96, 64, 0, False, []),
    "mix1_full": ("mix1", 0, 0, 96, 64, 0, False, []),
    "mix2_full": ("mix2", 0, 0, 96, 64, 0, False, []),
    "mix0_tile": ("mix0", 30, 20, 16, 16, 0, True, []),
    "mesh0_full": ("mesh0", 0, 0, 96, 64, 0, False, []),            # OBJ ingest through `model`
    "inst1_full": ("inst1", 0, 0, 112, 72, 0, False, []),           # grouping corner cases: mixed shared child, nested wrappers, bare + placed, single placement
    "inst2_full": ("inst2", 0, 0, 128, 80, 0, False, []),           # a composite prefab x 6 (mesh inside the prefab: a group inside a group, sphere, placed cube, pedestal) + the mesh by itself
    "inst0_full": ("inst0", 0, 0, 128, 80, 0, False, []),           # one mesh placed 36 times + one placed 8 times (two-level BVH on the device)
    # the reference's own demo workload (scene_management.hpp:103-236): 900 instanced prefabs, a wrapped mesh, fog
    "demo_tile": ("demo", 560, 300, 48, 32, 32, False, []),          # mesh + mirror sphere + glass cube
    "demo_tile_b": ("demo", 200, 420, 40, 24, 32, True, []),         # the instance grid through the fog
    "cfg3w_small": ("cfg3w", 900, 500, 24, 24, 64, False, [200, 20, 256, 128]),
}
TRACES = {
    # name -> (scene, rays, seed, probe-box clamp lo, hi, extra scene args)
    "trace_mix0": ("mix0", 4096, 11, -6, 6, []),
    "trace_cfg2": ("cfg2", 2048, 12, -12, 12, []),
    "trace_cfg5": ("cfg5", 2048, 13, -30, 600, []),
    "trace_cfg3_small": ("cfg3", 2048, 14, -4, 4, [200, 20, 256, 128]),
    "trace_mesh0": ("mesh0", 4096, 15, -3, 4, []),
    "trace_inst0": ("inst0", 4096, 18, -7, 7, []),
    "trace_inst1": ("inst1", 4096, 19, -5, 5, []),
    "trace_inst2": ("inst2", 4096, 24, -6, 6, []),
    "trace_demo": ("demo", 4096, 16, -16, 16, []),
    "trace_cfg3w_small": ("cfg3w", 4096, 17, -4, 4, [200, 20, 256, 128]),   # a placed (wrapped) mesh: baked to world space on the device
    # adversarial rays (ZR_TRACE_ADVERSARIAL: zero direction components, tiny/large scales, round and far origins)
    "trace_adv_mix0": ("mix0", 4096, 21, -6, 6, [], True),
    "trace_adv_cfg2": ("cfg2", 4096, 22, -12, 12, [], True),
    "trace_adv_cfg3_small": ("cfg3", 4096, 23, -4, 4, [200, 20, 256, 128], True),
}


# first-hit AOV tiles: name -> (scene, x0, y0, w, h, z_depth_max_dist, extra scene args)
AOVS = {
    "aov_mix0": ("mix0", 0, 0, 96, 64, 20.0, []),
    "aov_mix1": ("mix1", 8, 8, 64, 40, 12.0, []),
    "aov_cfg2": ("cfg2", 560, 280, 48, 32, 20.0, []),
    "aov_mesh0": ("mesh0", 0, 0, 96, 64, 9.0, []),
    "aov_inst0": ("inst0", 0, 0, 128, 80, 25.0, []),
    "aov_inst2": ("inst2", 0, 0, 128, 80, 25.0, []),
}

# beauty / reflection / refraction with the split flags on (camera.hpp:490-517): (scene, x0, y0, w, h, spp, scene args)
PASSES = {
    "passes_mix0": ("mix0", 0, 0, 96, 64, 16, []),
    "passes_mix2": ("mix2", 0, 0, 64, 48, 16, []),
    "passes_cfg2": ("cfg2", 560, 280, 48, 32, 32, []),
    "passes_cfg5": ("cfg5", 150, 350, 32, 24, 32, []),
    "passes_inst0": ("inst0", 16, 8, 96, 64, 16, []),
}

# the post stack (bloom, sharpen, process, 8-bit; analyze_framebuffer) through the genuine post_processor / bloom_filter:
# (scene, spp of the input frame, presets of `zenith_ref post`)
POSTS = {
    "post_cfg1": ("cfg1", 4, [0, 1, 2, 3, 4, 5, 6, 7]),
    "post_mix0": ("mix0", 8, [1, 3]),
}


def kat_inputs():
    """Hand-placed inputs of the per-function known-answer fixture (scene "kat0", scenes/zr_scenes_mix.inc).
    rays: (o, d, tmin, tmax); the comment names the reference behaviour each one pins."""
    INF = float("inf")
    R = []

    def ray(o, d, tmin=0.001, tmax=INF):
        R.append(list(o) + list(d) + [tmin, tmax])
    # sphere::hit (sphere.hpp:18-79): centre (0, 4, 0), r = 1, image-textured
    ray((0, 4, 5), (0, 0, -1))                       # near root, front face
    ray((0, 4, 0), (0, 0, -1))                       # origin inside: near root negative, far root taken, back face
    ray((0, 9, 0), (0, -1, 0))                       # north pole: up x n vanishes -> tangent fallback z x n (sphere.hpp:50-59), v = 1
    ray((0, 1, 0), (0, 1, 0))                        # south pole, v = 0
    ray((1e-9, 9, 0), (0, -1, 0))                    # a hair off the pole: the fallback threshold
    ray((1, 4, 5), (0, 0, -1))                       # tangent ray: discriminant exactly 0
    ray((0, 4, 5), (0, 0, -1), 0.001, 4.0)           # t == ray_t.max: surrounds() is strict -> miss
    ray((0, 4, 5), (0, 0, -1), 0.001, 4.000001)
    ray((0, 4, 5), (0, 0, -1), 4.0, INF)             # t == ray_t.min: strict -> the far root (t = 6)
    for k in range(8):                               # around the equator: u = (atan2(-z, x) + pi) / 2 pi incl. the seam
        c, sn = np.cos(k * np.pi / 4), np.sin(k * np.pi / 4)
        ray((5 * c, 4, 5 * sn), (-5 * c, 0, -5 * sn))          # un-normalised direction: t = 0.8
    ray((0, 4, 5), (0, 0, -1e-3))                    # short direction vector: t = 4000
    # cube::hit (cube.hpp:44-142): bare, origin-centred, half extents (1, 0.5, 0.75)
    ray((3, .1, .2), (-1, 0, 0)); ray((-3, .1, .2), (1, 0, 0)); ray((.3, 3, .2), (0, -1, 0))
    ray((.3, -2, .2), (0, 1, 0)); ray((.3, .1, 3), (0, 0, -1)); ray((.3, .1, -1.5), (0, 0, 1))
    ray((0.2, 0.1, -0.1), (1, 0.3, 0.2))             # origin inside: rec.t = ray_t.min, no face matches -> the +z fallthrough
    ray((3, 0.5, 0), (-1, 0, 0))                     # along the top face plane: 0 * inf = NaN in the y slab, fmax / fmin ignore it
    ray((3, 3, 3), (-1, -1, -1))
    ray((3, 0.5000001, 0), (-1, 0, 0))               # just above the face (inside the padded box): miss
    # dielectric sphere (6, 0, 0)
    ray((6, 0, 5), (0, 0, -1))
    ray((6, 0, 0), (0, 0.6, 0.8))                    # from inside, normal incidence: exit IOR
    ray((6, 0.9, 0), (1, 0, 0))                      # from inside at 64 degrees: total internal reflection, no draw
    ray((6.95, 0, 5), (0, 0, -1))                    # grazing from outside: Schlick close to 1
    # metal
    ray((0, -4, 5), (0, 0, -1))                      # fuzz 0 still draws a unit vector (material.hpp:141)
    ray((2, -4, 5), (-0.3, 0, -1))
    ray((6.99, 4, 5), (0, 0, -1))                    # fuzz 0.4 at grazing incidence: scatter may return false
    ray((6, 4, 5), (0, 0, -1))                       # bump map + nested checker
    ray((-6, 0, 5), (0, 0, -1))                      # diffuse_light: emitted = texture, scatter false
    ray((-6, 3, 5), (0, 0, -1))                      # failed image load: cyan
    ray((-6.1, -3, 5), (0, 0, -1))                   # sphere built with r = -0.75: box from |r|, hit radius max(0, r) = 0 -> miss
    ray((6, -4, 5), (0, 0, -1)); ray((6.5, -4.3, 5), (0, 0, -1))      # bumped lambertian
    # triangle::hit (triangle.hpp:17-82): A (3,0,-2) B (5,0,-2) C (3,2,-2), plane z = -2
    ray((3.5, 0.5, 3), (0, 0, -1))
    ray((4, 0, 3), (0, 0, -1))                       # on edge AB: the edge test is >= 0
    ray((3, 0, 3), (0, 0, -1))                       # on vertex A
    ray((4, 1, 3), (0, 0, -1))                       # on edge BC
    ray((4, -1e-12, 3), (0, 0, -1))                  # a hair outside
    ray((3.5, 0.5, 3), (0, 0, -1), 0.001, 5.0)       # t == ray_t.max: contains() is inclusive -> hit
    ray((3.5, 0.5, 3), (0, 0, -1), 0.001, 4.999999999)
    ray((3.5, 0.5, 3), (0, 0, -1), 5.0, INF)         # t == ray_t.min: inclusive -> hit
    ray((3.5, 0.5, -2), (1, 0, 0))                   # in the plane: |N.d| < 1e-8 -> miss
    ray((3.5, 0.5, -5), (0, 0, 1))                   # back face
    ray((3.0, 0.5, -1.9999999999), (1, 0, -1e-9))    # geometrically a hit, rejected by the parallel threshold
    ray((3.5, 0.5, 3), (0, 0, -1e6))                 # t = 5e-6 < ray_t.min
    ray((1.5, 9, 0), (0, -1, 0))                     # through the degenerate triangle at y = 8: never hit
    ray((-4, 0.8333, 3), (0, 0, -1)); ray((-4, 0.8333, -6), (0, 0, 1))   # tilted triangle with smooth normals, both sides
    # cube under translate: front_face forced true (translate.hpp:29)
    ray((0.1, 0.2, 9), (0, 0, -1)); ray((0, 0, 6), (0, 0, 1))
    # constant_medium in a sphere (0, 0, -8) r = 2
    ray((0, 0, -3), (0, 0, -1)); ray((0, 0, -8), (1, 0, 0)); ray((3, 0, -3), (0, 0, -1)); ray((0, 0, -3), (0, 0, -1), 0.001, 4.0)
    # closest of several, signed zeros, far origin
    ray((-10, 0, 0), (1, 0, 0)); ray((10, 0, 0), (-1, 0, 0)); ray((0, 4, 5), (-0.0, 0.0, -1)); ray((0, 4, 1e6), (0, 0, -1))
    rays = np.array(R, dtype=np.float64)

    # texture::value: every kat texture at and beyond the edges of the unit square
    edge = [0.0, 1.0, -0.25, 1.25, 0.5, 0.999999, 1e-9, 2.5, -3.75, 0.0625, 0.9375]
    pts = [(0, 0, 0), (0.5, 0, 0), (-0.5, 0.25, 1.5), (1.4999999, -1.5, 0.7), (-1e-12, 3.0, -0.75)]
    T = [[t, u, v, *pts[(iu + 3 * iv + t) % len(pts)]] for t in range(6) for iu, u in enumerate(edge) for iv, v in enumerate(edge)]
    tex = np.array(T, dtype=np.float64)

    # get_background_color: mode, bg rgb, intensity, yaw, tilt, roll, sun dir, sun colour, sun intensity, sun size | direction
    s0 = np.array([1.0, 0.5, -0.5]) / np.linalg.norm([1.0, 0.5, -0.5])
    envs = [
        [0, 0, 0, 0, 1.0, 0, 0, 0, *s0, 1, 1, 1, 1.0, 1.0],                       # the default PHYSICAL_SUN
        [0, 0, 0, 0, 0.8, 0, 0, 0, 1.0, 0.06, -0.45, 1, 0.9, 0.8, 3.0, 8.0],      # low sun: sunset branch, not normalised
        [0, 0, 0, 0, 1.3, 0, 0, 0, 1.0, -0.02, -0.3, 1, 1, 1, 2.0, 4.0],          # sun just below the horizon
        [0, 0, 0, 0, 1.0, 0, 0, 0, 1.0, -0.5, 0.0, 1, 1, 1, 1.0, 1.0],            # night: no disc
        [2, 0.25, 0.3, 0.35, 1.2, 0, 0, 0, *s0, 1, 1, 1, 1.0, 1.0],               # SOLID_COLOR
        [1, 0, 0, 0, 0.75, 1.1, 0.35, -0.4, *s0, 1, 1, 1, 1.0, 1.0],              # HDR_MAP, yaw / tilt / roll
        [1, 0, 0, 0, 1.0, 0, 0, 0, *s0, 1, 1, 1, 1.0, 1.0],                       # HDR_MAP, no rotation: poles and the seam
    ]
    rng = np.random.default_rng(20261004)
    dirs = [(0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1), (1, 1e-12, 0), (1, -1e-12, 0), (-1, 0, 1e-12), (-1, 0, -1e-12),
            (3, 4, 0), (0.001, 0.002, 0.002)]
    dirs += [tuple(v) for v in rng.normal(size=(20, 3))]
    B = []
    for e in envs:
        sun = np.array(e[8:11]) / np.linalg.norm(e[8:11])
        mine = list(dirs)
        if e[0] == 0:   # at, inside and across the edge of the sun disc (threshold 1 - sun_size / 1000, smoothstep 2e-4 wide)
            thr = 1.0 - e[15] * 0.001
            ortho = np.cross(sun, [0.3, 0.1, 0.9]); ortho /= np.linalg.norm(ortho)
            for cosang in (1.0, thr + 0.0003, thr + 0.0001, thr + 0.00005, thr - 0.00001):
                mine.append(tuple(cosang * sun + np.sqrt(max(0.0, 1 - cosang * cosang)) * ortho))
        B += [e + list(d) for d in mine]
    bg = np.array(B, dtype=np.float64)
    cam = {"kat0": np.array([[i, j, s] for (i, j) in [(0, 0), (63, 0), (0, 47), (63, 47), (31, 23)] for s in range(4)], dtype=np.float64),
           "cfg1": np.array([[i, j, s] for (i, j) in [(0, 0), (399, 0), (0, 224), (399, 224), (200, 112)] for s in range(2)], dtype=np.float64)}
    return rays, tex, bg, cam


def make_kat(tmp):
    rays, tex, bg, cam = kat_inputs()
    arrays, metas = {"rays": rays, "tex_in": tex, "bg_in": bg}, {}
    for what, arr, outs in (("hits", rays, ("recs", "scat")), ("tex", tex, ("rgb",)), ("bg", bg, ("rgb",))):
        f = os.path.join(tmp, f"kat_{what}.bin"); pre = os.path.join(tmp, f"kat_{what}")
        arr.tofile(f)
        metas[what] = run("kat", "kat0", what, f, len(arr), pre)
        for o in outs:
            arrays[f"{what}_{o}" if what != "hits" else o] = np.load(f"{pre}_{o}.npy")
    for scene, req in cam.items():
        f = os.path.join(tmp, f"kat_cam_{scene}.bin"); pre = os.path.join(tmp, f"kat_cam_{scene}")
        req.tofile(f)
        metas[f"cam_{scene}"] = run("kat", scene, "cam", f, len(req), pre)
        arrays[f"cam_req_{scene}"] = req.astype(np.int32)
        arrays[f"cam_rays_{scene}"] = np.load(pre + "_rays.npy")
    np.savez_compressed(os.path.join(HERE, "kat_kat0.npz"), meta=np.array(json.dumps(metas)), **arrays)
    print("kat_kat0", {k: v.shape for k, v in arrays.items()})


def make_refdemo(tmp):
    """The reference's OWN demo world (load_materials + sceneAssetsLoader + build_geometry, scene_management.hpp:28-236) for the GPU: the scene reads the
    reference's assets/ at run time, which never travel — so the world travels FLATTENED.  oracle/_ref/refscene_dropin (the reference's scene code compiled
    unchanged against the drop-in) dumps every array of the zr_scene_desc it flattens to (decoded texels included), oracle/_ref/zenith_ref (the same scene
    code against the reference's own headers) renders the full frame at 8 spp and two tiles at the scene's 16 spp.  tests/test_refdemo_gpu.py commits the
    arrays through the C ABI and compares."""
    from refdemo_assets import asset_dir
    dropin = os.path.join(ROOT, "oracle", "_ref", "refscene_dropin")
    os.makedirs(os.path.join(tmp, "refdemo_cwd"))
    cwd = asset_dir(os.path.join(tmp, "refdemo_cwd"))
    pre = os.path.join(tmp, "refdemo_d")
    p = subprocess.run([dropin, "dump", pre], cwd=cwd, capture_output=True, text=True, check=True)
    meta = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    names = ["spheres", "sphere_mat", "tri_v", "tri_n", "tri_mat", "cubes", "cube_mat", "media", "ops", "objects", "groups", "materials", "textures", "texels", "camera", "env"]
    arrays = {n: np.load(f"{pre}_{n}.npy") for n in names}
    np.savez_compressed(os.path.join(HERE, "refdemo_scene.npz"), meta=np.array(json.dumps(meta)), **arrays)
    print("refdemo_scene", meta)
    frames = {}
    for name, (x0, y0, w, h, spp) in {"full8": (0, 0, 640, 360, 8), "tile_a": (40, 250, 24, 16, 16), "tile_b": (300, 150, 32, 24, 16)}.items():
        out = os.path.join(tmp, "refdemo_" + name)
        r = subprocess.run([REF, "tile", "refdemo", str(x0), str(y0), str(w), str(h), str(spp), str(os.cpu_count() or 1), out, "0"], cwd=cwd, capture_output=True, text=True, check=True)
        m = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        frames[name] = np.load(out + "_mean.npy"); frames[name + "_meta"] = np.array(json.dumps(m))
        print("refdemo", name, m)
    np.savez_compressed(os.path.join(HERE, "refdemo_frames.npz"), **frames)


def run(*args):
    p = subprocess.run([REF] + [str(a) for a in args], capture_output=True, text=True, check=True)
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/zenith_ref missing: run `make -C oracle ref` (needs /root/reference)")
    want = set(sys.argv[1:])
    with tempfile.TemporaryDirectory() as tmp:
        for name, (scene, x0, y0, w, h, spp, ps, extra) in TILES.items():
            if want and name not in want and scene not in want:
                continue
            pre = os.path.join(tmp, name)
            meta = run("tile", scene, x0, y0, w, h, spp, os.cpu_count() or 1, pre, 1 if ps else 0, *extra)
            meta["scene_args"] = extra
            arrays = {"mean": np.load(pre + "_mean.npy"), "meta": np.array(json.dumps(meta))}
            if ps:
                arrays["samples"] = np.load(pre + "_samples.npy")
                arrays["counts"] = np.load(pre + "_counts.npy")
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
            print(name, meta)
        for name, spec in TRACES.items():
            scene, n, seed, clo, chi, extra = spec[:6]
            adv = len(spec) > 6 and spec[6]
            if want and name not in want and scene not in want:
                continue
            pre = os.path.join(tmp, name)
            if adv:
                os.environ["ZR_TRACE_ADVERSARIAL"] = "1"
            else:
                os.environ.pop("ZR_TRACE_ADVERSARIAL", None)
            meta = run("trace", scene, n, seed, pre, clo, chi, *extra)
            os.environ.pop("ZR_TRACE_ADVERSARIAL", None)
            meta["scene_args"] = extra
            meta["adversarial"] = bool(adv)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), rays=np.load(pre + "_rays.npy"),
                                recs=np.load(pre + "_recs.npy"), meta=np.array(json.dumps(meta)))
            print(name, meta)
        for name, (scene, x0, y0, w, h, zmax, extra) in AOVS.items():
            if want and name not in want and scene not in want:
                continue
            pre = os.path.join(tmp, name)
            meta = run("aov", scene, x0, y0, w, h, zmax, pre, "-", *extra)
            meta["scene_args"] = extra
            np.savez_compressed(os.path.join(HERE, name + ".npz"), albedo=np.load(pre + "_albedo.npy"), normal=np.load(pre + "_normal.npy"),
                                zdepth=np.load(pre + "_zdepth.npy"), meta=np.array(json.dumps(meta)))
            print(name, meta)
        for name, (scene, x0, y0, w, h, spp, extra) in PASSES.items():
            if want and name not in want and scene not in want:
                continue
            pre = os.path.join(tmp, name)
            meta = run("passes", scene, x0, y0, w, h, spp, pre, *extra)
            meta["scene_args"] = extra
            np.savez_compressed(os.path.join(HERE, name + ".npz"), beauty=np.load(pre + "_beauty.npy"), reflection=np.load(pre + "_reflection.npy"),
                                refraction=np.load(pre + "_refraction.npy"), meta=np.array(json.dumps(meta)))
            print(name, meta)
        for name, (scene, spp, presets) in POSTS.items():
            if want and name not in want and scene not in want:
                continue
            arrays, metas = {}, []
            for k in presets:
                pre = os.path.join(tmp, f"{name}_{k}")
                meta = run("post", scene, k, pre, spp)
                metas.append(meta)
                arrays["frame"] = np.load(pre + "_frame.npy")      # the same input frame for every preset
                arrays[f"rgb8_{k}"] = np.load(pre + "_rgb8.npy")
                arrays["hist"] = np.load(pre + "_hist.npy")
            np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=np.array(json.dumps(metas)), **arrays)
            print(name, [m["preset"] for m in metas])
        if not want or "refdemo" in want:
            make_refdemo(tmp)
        if not want or "kat" in want or "kat0" in want:
            make_kat(tmp)
        if not want or "texels" in want:
            out = os.path.join(tmp, "texels.npy")
            subprocess.run([REF, "texels", "64", "32", out], check=True)
            np.savez_compressed(os.path.join(HERE, "texels_hdr_64x32.npz"), texels=np.load(out))
            print("texels 64x32")


if __name__ == "__main__":
    main()
