"""The N > 1 path on CPU: two gloo ranks render interleaved pixel tiles of one frame and sum-reduce the double3
accumulator to rank 0 (raytracer_project_amd/multi.py — the same driver bench.py runs over RCCL).  On CPU the tile
renderer is the oracle (test infrastructure); on the GPU box tests/test_gpu_parity.py::test_tile_sharding_is_exact
covers the device side of the same property.  The reduced frame must equal the single-process frame bit for bit."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["ZR_ROOT"])
from raytracer_project_amd import capi, multi
from oracle import zr_oracle_py as zo

rank, local, world = multi.init_distributed(backend="gloo")
ds = capi.DemoScene("mix0")
cam = ds.camera.copy()
cam.samples_per_pixel = 4
osc = zo.OracleScene(ds.desc)
H, W = cam.image_height, cam.image_width
acc = torch.zeros((H, W, 3), dtype=torch.float64)

def render_tiles(r, w):
    region = multi.tile_region(capi, r, w, tile=16)
    out, _, _, _ = osc.render(cam, ds.env, ds.seed, region, threads=2)
    acc.copy_(torch.from_numpy(out))

multi.render_frame(render_tiles, acc, rank, world)
own = sum(1 for y in range(0, H, 16) for x in range(0, W, 16) if multi.owner_of_pixel(x, y, W, world, 16) == rank)
if rank == 0:
    np.save(os.environ["ZR_OUT"], acc.numpy())
print("rank", rank, "tiles", own)
dist.destroy_process_group()
'''


def test_two_rank_tile_sharding_gloo(built, tmp_path):
    from oracle import zr_oracle_py as zo
    from raytracer_project_amd import capi
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "frame.npy"
    env = dict(os.environ, ZR_ROOT=ROOT, ZR_OUT=str(out), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    tiles = sorted(int(l.split()[-1]) for l in p.stdout.splitlines() if l.startswith("rank"))
    assert tiles == [12, 12]  # 6 x 4 tiles of 16 px, interleaved
    ds = capi.DemoScene("mix0")
    cam = ds.camera.copy()
    cam.samples_per_pixel = 4
    full, _, _, _ = zo.OracleScene(ds.desc).render(cam, ds.env, ds.seed, None, threads=2)
    got = np.load(out)
    assert np.array_equal(got, full), "reduced multi-rank frame differs from the single-process frame"
