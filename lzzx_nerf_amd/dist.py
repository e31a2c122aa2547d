"""Multi-GPU: camera rays are independent, so a batch of rays is sharded contiguously across ranks (one process per
GPU), every rank renders its shard with a full replica of the model (tables 1.96 MB + MLPs 93 KB + bitfield 256 KB),
and the rendered tiles are all-gathered with RCCL over xGMI.  The reference has no reachable distributed code
(dead DDP wrapper only); this is new.

Payloads are tiny (512x512 frame = 3 MB of f32 RGB, 768 KB as RGB24), i.e. latency-bound on the 7 point-to-point
xGMI links: ONE all_gather per frame batch, uint8 when the consumer is the video pipe (TrainerUtil.py:550-555
quantises to uint8 before hand-off anyway), never per-iteration collectives.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:   # LZ_DIST_BACKEND=gloo: rehearse the multi-process path on a box with fewer GPUs than ranks
            backend = os.environ.get("LZ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" is RCCL on ROCm
        if torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", rank))
            torch.cuda.set_device(local % torch.cuda.device_count() if backend != "nccl" else local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_bounds(n_rays, rank, world):
    """contiguous, balanced split of n_rays: the first (n_rays % world) ranks get one extra ray"""
    base, rem = divmod(n_rays, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(rays_o, rays_d, rank, world):
    lo, hi = shard_bounds(rays_o.shape[0], rank, world)
    return rays_o[lo:hi].contiguous(), rays_d[lo:hi].contiguous()


def gather_tiles(tile, n_total=None, group=None):
    """all-gather per-rank tiles [n_local, C] (ragged allowed) into [n_total, C] on every rank, in rank order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tile
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    tile = tile.contiguous()
    if n_total is None or n_total % world == 0 and tile.shape[0] * world == n_total:
        out = torch.empty((tile.shape[0] * world,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        dist.all_gather_into_tensor(out, tile, group=group)
        return out
    # ragged: pad to the largest shard, gather once, strip
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    mx = max(h - l for l, h in sizes)
    pad = torch.zeros((mx,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    pad[: tile.shape[0]] = tile
    out = torch.empty((mx * world,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + (h - l)] for r, (l, h) in enumerate(sizes)], 0)


def to_rgb24(image):
    """[N,3] f32 in [0,1] -> uint8, the hand-off format of the reference's video pipe (TrainerUtil.py:550-555)"""
    return (image * 255).to(torch.uint8)
