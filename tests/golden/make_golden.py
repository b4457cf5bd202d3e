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
}

# beauty / reflection / refraction with the split flags on (camera.hpp:490-517): (scene, x0, y0, w, h, spp, scene args)
PASSES = {
    "passes_mix0": ("mix0", 0, 0, 96, 64, 16, []),
    "passes_mix2": ("mix2", 0, 0, 64, 48, 16, []),
    "passes_cfg2": ("cfg2", 560, 280, 48, 32, 32, []),
    "passes_cfg5": ("cfg5", 150, 350, 32, 24, 32, []),
}

# the post stack (bloom, sharpen, process, 8-bit; analyze_framebuffer) through the genuine post_processor / bloom_filter:
# (scene, spp of the input frame, presets of `zenith_ref post`)
POSTS = {
    "post_cfg1": ("cfg1", 4, [0, 1, 2, 3, 4, 5, 6, 7]),
    "post_mix0": ("mix0", 8, [1, 3]),
}


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
        if not want or "texels" in want:
            out = os.path.join(tmp, "texels.npy")
            subprocess.run([REF, "texels", "64", "32", out], check=True)
            np.savez_compressed(os.path.join(HERE, "texels_hdr_64x32.npz"), texels=np.load(out))
            print("texels 64x32")


if __name__ == "__main__":
    main()
