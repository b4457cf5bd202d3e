"""The N > 1 path on CPU: two gloo ranks render interleaved pixel tiles of one frame and exchange them — the default
packed-tile gather onto rank 0 and the whole-frame sum-reduce — so that rank 0 holds the frame (raytracer_project_amd/multi.py — the same driver bench.py runs over RCCL).  On CPU the tile
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

TILE = int(os.environ["ZR_TILE"])
rank, local, world = multi.init_distributed(backend="gloo")
ds = capi.DemoScene("mix0")
cam = ds.camera.copy()
cam.samples_per_pixel = 4
osc = zo.OracleScene(ds.desc)
H, W = cam.image_height, cam.image_width
acc = torch.zeros((H, W, 3), dtype=torch.float64)

def render_tiles(r, w):
    region = multi.tile_region(capi, r, w, tile=TILE)
    out, _, _, _ = osc.render(cam, ds.env, ds.seed, region, threads=2)
    acc.copy_(torch.from_numpy(out))

multi.render_frame(render_tiles, acc, rank, world, tile=TILE)
own = sum(1 for y in range(0, H, TILE) for x in range(0, W, TILE) if multi.owner_of_pixel(x, y, W, world, TILE) == rank)
if rank == 0:
    np.save(os.environ["ZR_OUT"], acc.numpy())
print("rank", rank, "tiles", own)
dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("exchange,world,tile", [("gather", 2, 16), ("reduce", 2, 16), ("gather", 3, 20)])
def test_tile_sharding_gloo(exchange, world, tile, built, tmp_path):
    """(gather, 3 ranks, 20-pixel tiles): 5 x 4 tiles with clipped ones at the right and bottom edges, shares of unequal size —
    the padded part of a share must not leak into the frame"""
    from oracle import zr_oracle_py as zo
    from raytracer_project_amd import capi
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "frame.npy"
    env = dict(os.environ, ZR_ROOT=ROOT, ZR_OUT=str(out), OMP_NUM_THREADS="1", ZR_MULTI_EXCHANGE=exchange, ZR_TILE=str(tile))
    import socket
    for attempt in range(2):   # (the port is free when probed and may be taken by the time the launcher binds it: one retry on a fresh port)
        with socket.socket() as sk:   # a free port per case: back-to-back launches on one fixed port met its TIME_WAIT now and then
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(script)]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        lost_port = any(m in (p.stdout + p.stderr).lower() for m in ("address already in use", "eaddrinuse", "errno: 98"))
        if p.returncode == 0 or not lost_port:   # (only a lost rendezvous port is retried — a rank that crashed or disagreed fails the first time)
            break
    assert p.returncode == 0, p.stdout + p.stderr
    tiles = sorted(int(l.split()[-1]) for l in p.stdout.splitlines() if l.startswith("rank"))
    n_tiles = ((96 + tile - 1) // tile) * ((64 + tile - 1) // tile)   # mix0 is 96 x 64
    assert sum(tiles) == n_tiles and tiles[-1] - tiles[0] <= 1, tiles
    ds = capi.DemoScene("mix0")
    cam = ds.camera.copy()
    cam.samples_per_pixel = 4
    full, _, _, _ = zo.OracleScene(ds.desc).render(cam, ds.env, ds.seed, None, threads=2)
    got = np.load(out)
    assert np.array_equal(got, full), "reduced multi-rank frame differs from the single-process frame"
