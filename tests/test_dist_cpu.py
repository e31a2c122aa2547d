"""N > 1 path on CPU: world_size 2 and 3 over gloo -- ray sharding + ONE all-gather of rendered tiles must reproduce
the single-process result exactly (rays are independent; there is no data-path collective besides the gather)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lzzx_nerf_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(rays_o, rays_d):
    """stand-in for the per-ray renderer: any function that treats rays independently"""
    return torch.sigmoid(rays_o * 0.3 + rays_d.flip(-1) * 2.0 + (rays_o * rays_d).sum(-1, keepdim=True))


def _worker(rank, world, port, n_rays, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    rays_o, rays_d = torch.randn(n_rays, 3, generator=g), torch.randn(n_rays, 3, generator=g)
    ro, rd = D.shard_rays(rays_o, rays_d, rank, world)
    lo, hi = D.shard_bounds(n_rays, rank, world)
    assert ro.shape[0] == hi - lo
    tile = _fake_render(ro, rd)
    full = D.gather_tiles(tile, n_total=n_rays)
    rgb24 = D.gather_tiles(D.to_rgb24(tile), n_total=n_rays)
    ref = _fake_render(rays_o, rays_d)
    ok = torch.equal(full, ref) and torch.equal(rgb24, D.to_rgb24(ref))
    # weak-scaling arrangement of bench.py: rank r renders "frame r", the gathered batch is frame-major
    frame = _fake_render(rays_o + rank, rays_d)
    batch = D.gather_tiles(frame)
    ok = ok and batch.shape[0] == world * n_rays and torch.equal(batch[rank * n_rays:(rank + 1) * n_rays], frame)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # the max-over-ranks timing reduction of bench.py
    ok = ok and float(t) == float(world)
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rays", [(2, 4096), (2, 4097), (3, 1000)])
def test_ray_sharding_and_tile_gather(tmp_path, world, n_rays):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rays, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert int(np.load(tmp_path / f"ok_{r}.npy")[0]) == 1


def test_shard_bounds_partition():
    for n in (0, 1, 7, 262144, 262145):
        for w in (1, 2, 3, 8):
            b = [D.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [h - l for l, h in b]
            assert max(sizes) - min(sizes) <= 1
