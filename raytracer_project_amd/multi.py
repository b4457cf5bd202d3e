"""Pixel-tile sharding of one frame across the GPUs of a node: one process per GPU, scene replicated,
interleaved 32x32 tiles (tile t -> rank t % world), and ONE collective per frame — a sum-reduce of the
zero-initialised double3 accumulator to rank 0 (torch.distributed backend "nccl" = RCCL over xGMI).

The reference shards rows across std::threads with no communication (camera.hpp:557-573); pixels are
independent given the counter RNG (include/zr_rng.h), so there is no data-path collective other than the final
gather.  Tiles are disjoint, so every pixel is the sum of one value and world-1 zeros: the reduce is exact and
the multi-GPU image is bit-identical to the single-GPU one (tests/test_multi_gloo.py, tests/test_gpu_parity.py).

`render_tiles` is injected so that the same driver is exercised on CPU with gloo (tests) and on GPUs with RCCL
(bench.py); this module contains no rendering arithmetic."""
import os

TILE = 32


def tile_region(capi, rank, world, tile=TILE):
    """The zr_region selecting this rank's tiles of the full frame."""
    return capi.Region(0, 0, 0, 0, tile, world if world > 1 else 0, rank if world > 1 else 0, 0)


def owner_of_pixel(x, y, width, world, tile=TILE):
    tiles_x = (width + tile - 1) // tile
    return ((y // tile) * tiles_x + (x // tile)) % world


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run).  Returns (rank, local_rank, world)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("ZR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def reduce_frame(acc, world, dst=0):
    """The one collective of a frame: sum the per-rank accumulators (disjoint tiles) onto rank `dst`.
    With the gloo backend (rehearsals on a box without one GPU per rank) device tensors are staged through the host."""
    if world <= 1:
        return acc
    import torch.distributed as dist
    if dist.get_backend() == "gloo" and acc.is_cuda:
        host = acc.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            acc.copy_(host)
        return acc
    dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
    return acc


def all_reduce_values(values, world, device, op="sum"):
    """all-reduce a short list of Python floats (bookkeeping: segment counts, elapsed time)."""
    import torch
    if world <= 1:
        return list(values)
    import torch.distributed as dist
    dev = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return t.tolist()


def render_frame(render_tiles, acc, rank, world):
    """acc: zero-initialised (H, W, 3) float64 tensor on this rank's device.  render_tiles(region_rank, region_world)
    must write this rank's pixels into acc (stream-ordered before the collective).  After the call rank 0 holds the
    whole frame."""
    render_tiles(rank, world)
    return reduce_frame(acc, world)
