"""Pixel-tile sharding of one frame across the GPUs of a node: one process per GPU, scene replicated,
interleaved 32x32 tiles (tile (tx, ty) -> rank (tx + skew * ty) % world: `tile_skew`), and ONE collective per frame (torch.distributed backend
"nccl" = RCCL over xGMI): every rank packs the pixels of ITS tiles (1/world of the frame) and one gather hands
the packed tiles to rank 0, which scatters them into the frame.  Each rank therefore sends 1/world of the double3 frame
over each of its xGMI links (6.2 MB at 1080p and 8 ranks) instead of pushing the whole 49.8 MB frame — 7/8 of it
zeros — through a ring sum-reduce (`exchange="reduce"`, the round-1 form, is kept for comparison).

The reference shards rows across std::threads with no communication (camera.hpp:557-573); pixels are
independent given the counter RNG (include/zr_rng.h), so there is no data-path collective other than this final
exchange, and it moves values without arithmetic: the multi-GPU image is bit-identical to the single-GPU one
(tests/test_multi_gloo.py, tests/test_gpu_parity.py).  north_star speaks of a "reduce of the float3 accumulator"; the
accumulator here is the reference's double3 render_accumulator (camera.hpp:420) and stays double — a float3 view would
halve 6 MB of traffic that is already negligible and cost the bit-exactness.

`render_tiles` is injected so that the same driver is exercised on CPU with gloo (tests) and on GPUs with RCCL
(bench.py); this module contains no rendering arithmetic."""
import os

# side of the interleaved pixel tiles (tile t -> rank t % world).  32 is the library's default tile; ZR_MULTI_TILE overrides it for
# experiments (larger tiles keep a rank's rays closer together, at the price of coarser load balance)
TILE = int(os.environ.get("ZR_MULTI_TILE", "32"))


def tile_skew(world):
    """The lattice the tiles are dealt on: tile (tx, ty) -> rank (tx + skew * ty) % world (zr_region::tile_skew).  The row-major rule t % world of rounds 1-3 gives
    near-vertical stripes whenever the tiles per row are a multiple of world / 2 (1080p: 60 tiles per row, 8 ranks — a rank owned two tile columns of every
    row, and the ranks under the middle of the picture were 4 % slower than the others); a skew near 0.38 * world that is coprime to world spreads every
    rank's tiles over all columns and rows.  ZR_MULTI_SKEW overrides (0 = the old rule)."""
    env = os.environ.get("ZR_MULTI_SKEW")
    if env is not None:
        return max(0, int(env))
    if world <= 2:
        return 1 if world == 2 else 0
    from math import gcd
    s = max(1, round(0.382 * world))
    while gcd(s, world) != 1:
        s += 1
    return s


def tile_region(capi, rank, world, tile=TILE):
    """The zr_region selecting this rank's tiles of the full frame."""
    return capi.Region(0, 0, 0, 0, tile, world if world > 1 else 0, rank if world > 1 else 0, tile_skew(world) if world > 1 else 0)


def owner_of_pixel(x, y, width, world, tile=TILE):
    tiles_x = (width + tile - 1) // tile
    skew = tile_skew(world)
    if skew > 0:
        return ((x // tile) + skew * (y // tile)) % world
    return ((y // tile) * tiles_x + (x // tile)) % world


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run).  Returns (rank, local_rank, world)."""
    # dmabuf IPC: RCCL between processes (and CUDA-tensor sharing) needs it on this host driver, and it must be in the environment before the first
    # HIP call of the process — a launcher that starts the ranks itself (torch.distributed.run by the driver) never passes through bench.self_launch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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


def reduce_frame(acc, world, dst=0, _force=False):
    """The one collective of a frame: sum the per-rank accumulators (disjoint tiles) onto rank `dst`.
    With the gloo backend (rehearsals on a box without one GPU per rank) device tensors are staged through the host.
    `_force` (tests): run the collective even in a world of one, so that a one-GPU box executes `dist.reduce` on RCCL."""
    if world <= 1 and not _force:
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


_owned = {}


def owned_pixels(height, width, world, device, tile=TILE):
    """per rank: flat pixel indices (y * width + x, ascending) of the tiles it owns; cached"""
    import torch
    skew = tile_skew(world)
    key = (height, width, world, str(device), tile, skew)
    if key not in _owned:
        ys = torch.arange(height, device=device).view(-1, 1)
        xs = torch.arange(width, device=device).view(1, -1)
        tiles_x = (width + tile - 1) // tile
        ty, tx = torch.div(ys, tile, rounding_mode="floor"), torch.div(xs, tile, rounding_mode="floor")
        owner = ((tx + skew * ty) if skew > 0 else (ty * tiles_x + tx)) % world
        _owned[key] = [torch.nonzero(owner.reshape(-1) == r).reshape(-1) for r in range(world)]
    return _owned[key]


def gather_frame(acc, world, dst=0, tile=TILE, _force=False):
    """The one collective of a frame: a GATHER of every rank's packed tiles (1/world of the frame each, padded to the
    largest share) onto rank `dst`, which scatters them into `acc`.  Only `dst` needs the frame, so only `dst` receives:
    each share crosses one xGMI link once (round 2 used an all-gather, which also delivered 7 shares to each of the 7 ranks
    that dropped them).  Pure data movement: bit-exact.
    `_force` (tests): run pack -> dist.gather -> scatter even in a world of one (the root then also scatters its own share back), so that a
    one-GPU box executes the exchange on the RCCL backend."""
    if world <= 1 and not _force:
        return acc
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    staged = dist.get_backend() == "gloo" and acc.is_cuda   # rehearsals on a box without one GPU per rank
    frame = acc.cpu() if staged else acc
    H, W = frame.shape[0], frame.shape[1]
    own = owned_pixels(H, W, world, frame.device, tile)
    m = max(int(o.numel()) for o in own)
    flat = frame.view(-1, 3)
    packed = torch.zeros((m, 3), dtype=frame.dtype, device=frame.device)
    packed[:own[rank].numel()] = flat[own[rank]]
    shares = [torch.empty((m, 3), dtype=frame.dtype, device=frame.device) for _ in range(world)] if rank == dst else None
    dist.gather(packed, shares, dst=dst)
    if rank == dst:
        if _force:
            flat[own[rank]] = float("nan")   # what comes back below can only have come through the collective
        for r in range(world):
            if r != rank or _force:
                flat[own[r]] = shares[r][:own[r].numel()]
        if staged:
            acc.copy_(frame)
    return acc


def exchange_frame(acc, world, dst=0, tile=TILE):
    """packed-tile gather onto `dst` by default; ZR_MULTI_EXCHANGE=reduce selects the whole-frame sum-reduce"""
    if os.environ.get("ZR_MULTI_EXCHANGE", "gather") == "reduce":
        return reduce_frame(acc, world, dst)
    return gather_frame(acc, world, dst, tile)


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


def render_frame(render_tiles, acc, rank, world, tile=TILE):
    """acc: zero-initialised (H, W, 3) float64 tensor on this rank's device.  render_tiles(region_rank, region_world)
    must write this rank's pixels into acc (stream-ordered before the collective).  After the call rank 0 holds the
    whole frame."""
    render_tiles(rank, world)
    return exchange_frame(acc, world, 0, tile)
