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


def _tiny_model():
    torch.manual_seed(3)
    table = torch.nn.Parameter(torch.rand(64, 2) * 2 - 1)
    mlp = torch.nn.Sequential(torch.nn.Linear(5, 16, bias=False), torch.nn.ReLU(), torch.nn.Linear(16, 3, bias=False))
    return table, mlp


def _tiny_loss(table, mlp, rays_o, rays_d, target):
    idx = (rays_o[:, 0].abs() * 1000).long() % table.shape[0]
    feat = torch.cat([table[idx], rays_d], -1)
    return ((torch.sigmoid(mlp(feat)) - target) ** 2).mean()


def _train_worker(rank, world, port, n_rays, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(1)
    rays_o, rays_d, target = torch.randn(n_rays, 3, generator=g), torch.randn(n_rays, 3, generator=g), torch.rand(n_rays, 3, generator=g)
    table, mlp = _tiny_model()
    if rank != 0:   # replicas start from rank 0's weights
        with torch.no_grad():
            for p in [table, *mlp.parameters()]:
                p.add_(1.0)
    D.broadcast_state([table.data, *[p.data for p in mlp.parameters()]])
    params = [table, *mlp.parameters()]
    bucket = D.GradientBucket(params)
    opt = torch.optim.Adam(params, lr=1e-2)
    lo, hi = D.shard_bounds(n_rays, rank, world)
    for _ in range(2):
        bucket.zero()
        _tiny_loss(table, mlp, rays_o[lo:hi], rays_d[lo:hi], target[lo:hi]).backward()
        bucket.all_reduce(weight=hi - lo)   # ragged shards: weight every rank's mean-normalised gradient by its ray count
        opt.step()
    # single-process full-batch reference on the same data
    rt, rm = _tiny_model()
    rp = [rt, *rm.parameters()]
    ropt = torch.optim.Adam(rp, lr=1e-2)
    for _ in range(2):
        ropt.zero_grad()
        _tiny_loss(rt, rm, rays_o, rays_d, target).backward()
        ropt.step()
    ok = all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(params, rp))
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    ok = ok and all(torch.equal(both[0], b) for b in both)   # replicas stay bit-identical
    opt.zero_grad(set_to_none=True)
    try:
        bucket.all_reduce()
        ok = False
    except RuntimeError:
        bucket.attach()
        bucket.all_reduce()
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rays", [(2, 2048), (3, 2050)])
def test_gradient_bucket_matches_full_batch(tmp_path, world, n_rays):
    """data-parallel training over (ragged) ray shards + ONE count-weighted all-reduce of the flat gradient buffer == full-batch training"""
    port = _free_port()
    mp.spawn(_train_worker, args=(world, port, n_rays, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert int(np.load(tmp_path / f"ok_{r}.npy")[0]) == 1


def test_gradient_bucket_single_process():
    table, mlp = _tiny_model()
    params = [table, *mlp.parameters()]
    bucket = D.GradientBucket(params)
    g = torch.Generator().manual_seed(1)
    _tiny_loss(table, mlp, torch.randn(64, 3, generator=g), torch.randn(64, 3, generator=g), torch.rand(64, 3, generator=g)).backward()
    assert bucket.flat.abs().sum() > 0 and all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in params)
    before = bucket.flat.clone()
    assert torch.equal(bucket.all_reduce(), before)
    bucket.zero()
    assert float(table.grad.abs().sum()) == 0.0


# ---- BASELINE cfg4 on CPU: one frame ray-sharded over the ranks through the product's ShardedFrame (tile lists, staging, ONE
# all-gather, re-assembly); the renderer under it is the CPU checker (the HIP renderer needs a GPU), on real rays of a real scene ----
def _frame_worker(rank, world, port, tiles, out_dir, via="collective"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      OMP_NUM_THREADS="2", OMP_WAIT_POLICY="passive")   # the checker's OpenMP teams would oversubscribe the host
    torch.set_num_threads(2)
    D.init_from_env(backend="gloo")
    from conftest import ellipsoid_bitfield
    from lzzx_nerf_amd.synthetic import load_golden, make_params, synthetic_camera
    from oracle import oracle as O
    from oracle.head import TriplaneSpec
    from oracle.render import render_inference
    golden = load_golden()
    P = make_params(golden)
    H, W = 40, 24          # 5 stripes of 8 rows: ragged for every world size used here
    pose, intr = synthetic_camera(H, W)
    bits, _ = ellipsoid_bitfield()
    render = lambda ro, rd: render_inference(TriplaneSpec(1.0), P, ro, rd, bits, golden["net_enc_a"], golden["net_ind"], golden["net_eye"],
                                             max_steps=24)["image"]
    sf = D.ShardedFrame(H, W, rank, world, tiles, device="cpu", via=via)
    assert sf.pixels.numel() == sf.n_local and sum(sf.sizes) == H * W
    if via == "peer":
        g = sf.gatherer
        assert isinstance(g, D.PeerTileGatherer) and g.offsets == [sum(sf.sizes[:r]) for r in range(world)] and g.total == H * W
    r = O.get_rays_batched(pose[None], intr, H, W, sf.pixels.numpy())
    ok = True
    for k in range(3):     # three frames through the double-buffered gatherer
        tile = torch.from_numpy(render(r["rays_o"][0], r["rays_d"][0])) + k
        frame = sf.assemble(sf.gather(tile))
        sf.wait()
        if k == 0:
            full = O.get_rays_batched(pose[None], intr, H, W)
            ref = torch.from_numpy(render(full["rays_o"][0], full["rays_d"][0]))
            ok = ok and float(ref.min()) < 0.99     # the ellipsoid is in view: not an all-background frame
        ok = ok and torch.equal(frame, ref + k)
        if via == "peer":   # every writer raised its flag in this frame's buffer; the other buffer still holds the previous frame
            g = sf.gatherer
            ok = ok and bool((g.flags[k & 1] == k + 1).all()) and int(g.timed_out) == 0
            if k >= 1:
                ok = ok and torch.equal(sf.assemble(g.frame[(k - 1) & 1]), ref + (k - 1))
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,tiles,via", [(2, "contiguous", "collective"), (2, "interleaved", "collective"), (3, "interleaved", "collective"),
                                             (2, "interleaved", "peer"), (3, "interleaved", "peer"), (3, "contiguous", "peer")])
def test_frame_ray_sharded_equals_unsharded(tmp_path, world, tiles, via):
    """via="peer": the collective-free hand-off (dist.PeerTileGatherer: every rank writes its tile at its offset of every peer's frame
    buffer, double-buffered, a flag per writer) -- on CPU the peer writes ride a gloo all_gather, so this pins the offsets / ragged
    sizes / buffer rotation / flag bookkeeping; the IPC transport itself needs GPUs (tests/test_gpu_cfg4.py rehearses it on one card)"""
    port = _free_port()
    mp.spawn(_frame_worker, args=(world, port, tiles, str(tmp_path), via), nprocs=world, join=True)
    for r in range(world):
        assert int(np.load(tmp_path / f"ok_{r}.npy")[0]) == 1


def test_tile_partitions():
    for tiles in ("contiguous", "interleaved"):
        for world in (1, 2, 3, 4, 8):
            for H, W in ((512, 512), (40, 24), (7, 5)):
                perm = D.frame_permutation(H, W, world, tiles)
                assert torch.equal(perm.sort().values, torch.arange(H * W))
                if tiles == "contiguous":
                    assert torch.equal(perm, torch.arange(H * W))
    assert D.tile_rows(512, 3, 8, "interleaved")[:9] == [24, 25, 26, 27, 28, 29, 30, 31, 88]


# ---- the reference's cap across ranks (round 4): tiles exchange ONE histogram and replay the same schedule ------------------------------
def _cap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_from_env(backend="gloo")
    from conftest import ellipsoid_bitfield, make_params, synthetic_camera
    from lzzx_nerf_amd.synthetic import load_golden
    from oracle.head import TriplaneSpec, get_rays
    from oracle.render import render_inference
    H = W = 32
    M = 16                                               # the reference's deployed max_steps: the cap binds
    golden = load_golden()
    P = make_params(golden)
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = ellipsoid_bitfield()[0]
    sf = D.ShardedFrame(H, W, rank, world, "interleaved", device="cpu")

    class R:                                             # what configure() sets on a TriplaneRenderer
        pass
    r = sf.configure(R())
    ok = r.cap == "reference" and r.frame_rays_total == H * W and r.cap_exchange == sf.sum_over_ranks
    px = sf.pixels.numpy()
    # Under the schedule n_step = 1 the alive count of iteration k IS the number of this tile's rays with L >= k: the histogram phase 1 builds
    st = {}
    render_inference(TriplaneSpec(1.0), P, ro[px], rd[px], bits, golden["net_enc_a"], golden["net_ind"], golden["net_eye"], max_steps=M, stats=st,
                     budget_factor=1, n_step_cap=1)
    alive = [n for n, _ in st["schedule"]] + [0] * (M + 1)
    hist = torch.tensor([alive[k] - alive[k + 1] for k in range(M)] + [0], dtype=torch.int32)
    hist[M] = len(px) - int(hist.sum())                  # everything else is still alive at the cap
    r.cap_exchange(hist)                                 # ONE all-reduce of max_steps + 1 words
    ok = ok and int(hist.sum()) == H * W
    c_eff, iters, bounds = D.cap_schedule_from_histogram(hist.tolist(), H * W, M)
    # the unsharded reference frame under the reference's own schedule
    full = {}
    render_inference(TriplaneSpec(1.0), P, ro, rd, bits, golden["net_enc_a"], golden["net_ind"], golden["net_eye"], max_steps=M, stats=full)
    ok = ok and c_eff == sum(n for _, n in full["schedule"]) and iters == len(full["schedule"]) and c_eff > M
    ok = ok and bounds[1:] == list(np.cumsum([n for _, n in full["schedule"]]))
    np.save(os.path.join(out_dir, f"cap_ok_{rank}.npy"), np.array([int(ok), c_eff, iters]))
    dist.barrier()
    dist.destroy_process_group()


def test_tiles_replay_the_reference_cap_from_one_summed_histogram(tmp_path):
    """world 2 over gloo: every rank histograms its tile's rays by their last surviving chunk boundary, ShardedFrame.sum_over_ranks (what
    configure() hands the fused renderer as cap_exchange) all-reduces the max_steps + 1 words, and the replay of the reference's n_step rule
    on the summed counts (dist.cap_schedule_from_histogram, the host mirror of lz_k_frame_schedule) gives every rank the C_eff, the
    iteration count and the chunk boundaries of the UNSHARDED frame under the reference's schedule -- with the CPU checker as the renderer"""
    port = _free_port()
    mp.spawn(_cap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [np.load(tmp_path / f"cap_ok_{r}.npy") for r in range(2)]
    assert all(int(g[0]) == 1 for g in got), got
    assert int(got[0][1]) == int(got[1][1]) > 16
