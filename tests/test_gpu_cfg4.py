"""BASELINE cfg4 on one GPU: the 512 x 512 frame rendered as 8 ray shards (both tilings) through the product's ShardedFrame +
TriplaneRenderer equals the unsharded frame bit for bit; bench.py's self-spawned 2-rank path (gloo on one card: a rehearsal of the
launch, sharding, gather and JSON contract -- not a scaling measurement) prints a well-formed strong-scaling line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tiles", ["contiguous", "interleaved"])
def test_frame_as_8_ray_shards_equals_unsharded(params, golden, tiles):
    from lzzx_nerf_amd import dist as D
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    from lzzx_nerf_amd.synthetic import ellipsoid_bitfield_device, synthetic_camera
    H = W = 512
    pose, intr = synthetic_camera(H, W)
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0)
    bits, _ = ellipsoid_bitfield_device("cuda")
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    r = TriplaneRenderer(head, bits, bound=1.0, budget_factor=4, n_step_cap=4)
    full = D.ShardedFrame(H, W, 0, 1, device="cuda")
    ro, rd = full.rays(dev(pose), intr)
    ref = r.render(ro, rd, *cond, max_steps=192, count_samples=True)
    ref_img, ref_cnt = ref["image"].clone(), ref["ray_counts"].clone()
    assert float(ref_img.min()) < 0.9                                   # the head is in view
    world = 8
    tiles_out, counts = [], []
    for g in range(world):
        sf = D.ShardedFrame(H, W, g, world, tiles, device="cuda")
        sf.gatherer = None                                              # one process: concatenate in rank order instead
        assert sf.n_local == H * W // world
        so, sd_ = sf.rays(dev(pose), intr)
        assert torch.equal(so, ro[sf.pixels]) and torch.equal(sd_, rd[sf.pixels])   # the tile's rays are the frame's rays at its pixels
        o = r.render(so, sd_, *cond, max_steps=192, count_samples=True)
        tiles_out.append(o["image"].clone())
        counts.append(o["ray_counts"].clone())
    frame = D.assemble_frame(torch.cat(tiles_out), H, W, world, tiles)
    cnt = D.assemble_frame(torch.cat(counts)[:, None], H, W, world, tiles)[:, 0]
    assert torch.equal(frame, ref_img)                                  # pixels: bit for bit
    assert torch.equal(cnt, ref_cnt)                                    # per-ray sample counts too
    if tiles == "interleaved":                                          # stripes balance the load: every shard marches a similar share
        per = torch.stack([c.sum() for c in counts]).double()
        assert float(per.max() / per.mean()) < 1.25


def test_pinned_schedule_makes_tiles_agree_on_rays_that_reach_the_cap(params, golden):
    """cap = "per_ray": fused mode stops a ray that is still alive at the cap after ceil(max_steps / S) * S samples (the reference tests the cap once per
    iteration, renderer.py:503-548) and picks S from the ray count -- so a tile and the whole frame differ on such rays unless S is
    pinned (ShardedFrame(steps_per_pass=)).  All-ones occupancy, dt_gamma 0 and a 19-step cap: every ray that enters the box hits it."""
    from lzzx_nerf_amd import dist as D
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    from lzzx_nerf_amd.synthetic import ones_bitfield, synthetic_camera
    H = W = 128
    pose, intr = synthetic_camera(H, W)
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0)
    bits = dev(ones_bitfield())
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    kw = dict(max_steps=19, dt_gamma=0.0, count_samples=True)
    world = 4
    for S, cap in ((1, 19), (4, 20), (8, 24)):
        full_sf = D.ShardedFrame(H, W, 0, 1, device="cuda", steps_per_pass=S, cap="per_ray")
        r = full_sf.configure(TriplaneRenderer(head, bits, bound=1.0, mode="fused"))
        ro, rd = full_sf.rays(dev(pose), intr)
        ref = {k: v.clone() for k, v in r.render(ro, rd, *cond, **kw).items()}
        assert int(ref["ray_counts"].max()) == cap and int((ref["ray_counts"] == cap).sum()) > 1000      # ceil(19 / S) * S
        loop = TriplaneRenderer(head, bits, bound=1.0, budget_factor=S, n_step_cap=S).render(ro, rd, *cond, **kw)
        assert torch.equal(loop["image"], ref["image"]) and torch.equal(loop["ray_counts"], ref["ray_counts"])
        imgs, cnts = [], []
        for g in range(world):
            sf = D.ShardedFrame(H, W, g, world, "interleaved", device="cuda", steps_per_pass=S, cap="per_ray")
            sf.gatherer = None
            rr = sf.configure(TriplaneRenderer(head, bits, bound=1.0, mode="fused"))
            o = rr.render(*sf.rays(dev(pose), intr), *cond, **kw)
            imgs.append(o["image"].clone())
            cnts.append(o["ray_counts"].clone())
        assert torch.equal(D.assemble_frame(torch.cat(imgs), H, W, world, "interleaved"), ref["image"]), S
        assert torch.equal(D.assemble_frame(torch.cat(cnts)[:, None], H, W, world, "interleaved")[:, 0], ref["ray_counts"]), S
    # unpinned: the 4 096-ray tile picks S > 1, the 16 384-ray frame may pick another -> counts at the cap differ, as documented
    auto_tile = TriplaneRenderer(head, bits, bound=1.0, mode="fused", cap="per_ray").render(*D.ShardedFrame(H, W, 0, world, "interleaved", device="cuda").rays(dev(pose), intr), *cond, **kw)
    assert int(auto_tile["ray_counts"].max()) in (19, 20, 24, 32)


def test_device_built_ellipsoid_bitfield_equals_checker():
    from conftest import ellipsoid_bitfield
    from lzzx_nerf_amd.synthetic import ellipsoid_bitfield_device
    bits, grid = ellipsoid_bitfield_device("cuda")
    bits_o, grid_o = ellipsoid_bitfield()
    assert np.array_equal(bits.cpu().numpy(), bits_o) and np.array_equal(grid.cpu().numpy(), grid_o)


@pytest.mark.parametrize("extra", [["--max-steps", "64", "--verify-frames", "3"], ["--max-steps", "16", "--scene", "ellipsoid"]])
def test_bench_self_spawns_two_ranks_and_prints_the_contract_line(tmp_path, extra):
    """(second case: the reference's deployed max_steps, where the cap binds -- the two ranks all-reduce the cap histogram between the two
    phases of every frame, and rank 0's check of the gathered frame against its own tile still holds)"""
    env = dict(os.environ, LZ_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "128"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["steps"] == 2 and r["unit"] == "samples/s" and r["value"] > 0
    assert r["gathered_frame_ok"] is True and r["gathered_frame_equals_unsharded_reference_loop"] is True
    assert "one frame ray-sharded x2" in r["config"]["parallelism"] and r["config"]["rays_per_rank"] == 128 * 128 // 2
    assert r["clip_weak_scaling"]["scaling"] == "weak" and r["clip_weak_scaling"]["frames_per_step"] == 2
    assert "tiles_contiguous" in r and "roofline" in r
    # the line the driver parses stays compact (21.9 KB was unparseable) and carries the contract's key set
    from tools.bench_contract import CONFIG_KEYS, CONTRACT_KEYS, MAX_LINE_BYTES, ROOFLINE_KEYS
    assert len(lines[0]) < MAX_LINE_BYTES // 2, len(lines[0])
    assert set(CONTRACT_KEYS) <= set(r) and set(CONFIG_KEYS) <= set(r["config"]) and set(ROOFLINE_KEYS) <= set(r["roofline"])
    assert p.stdout.strip().splitlines()[-1] == lines[0]
    assert len(r["rank_tile_ms"]["step"]) == 2 and len(r["rank_tile_ms"]["kernel"]) == 2 and min(r["rank_tile_ms"]["step"]) > 0
    if "--verify-frames" in extra:   # K more frames compared word for word on every rank, no barrier between them
        assert r["verify_frames"]["steps"] == 3 and r["verify_frames"]["mismatching_frames_over_all_ranks"] == 0
    detail = json.load(open(os.path.join(ROOT, r["detail"])))
    assert detail["value"] == r["value"] and "schedule" in detail["config"]


def test_peer_tile_hand_off_two_ranks_on_one_card(tmp_path):
    """--gather-via peer (dist.PeerTileGatherer): two processes on the one GPU of a test box export their frame buffers over HIP IPC, copy
    their tiles straight into each other's buffer, raise flags and wait for them on the device (lz_wait_flags) -- the whole transport
    except that both ends sit on the same card.  The gathered frame must equal each rank's own tile where they overlap, for every
    frame of the double-buffered sequence, and no wait may have timed out."""
    script = tmp_path / "peer2.py"
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["LZ_ROOT"])
from lzzx_nerf_amd import dist as D
rank, world = D.init_from_env(backend="gloo")
torch.cuda.set_device(0)
H, W = 64, 48
sf = D.ShardedFrame(H, W, rank, world, "interleaved", device="cuda", via="peer")
ok = True
for k in range(5):
    tile = torch.full((sf.n_local, 3), float(10 * k + rank), device="cuda") + sf.pixels[:, None].float() * 1e-3
    gathered = sf.gather(tile)
    sf.wait()
    frame = sf.assemble(gathered)
    torch.cuda.synchronize()
    for r in range(world):
        px = D.tile_pixels(H, W, r, world, "interleaved", "cuda")
        want = torch.full((px.numel(), 3), float(10 * k + r), device="cuda") + px[:, None].float() * 1e-3
        ok = ok and torch.equal(frame[px], want)
    dist.barrier()          # a rank may only overwrite a peer's buffer of frame k after the peer has read frame k - 2: keep the ranks within one frame
ok = ok and int(sf.gatherer.timed_out) == 0
print("peer ok" if ok else "peer MISMATCH", rank, flush=True)
dist.barrier()
dist.destroy_process_group()
''')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LZ_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "peer ok" in so, (so[-500:], se[-2000:])


def test_rccl_backend_runs_the_two_collectives_world_1(tmp_path):
    """One GPU is all a test box has, so the N-way exchange itself cannot run here; what can: the RCCL backend ("nccl" on ROCm) initialises
    on this image and executes, on the device and on a side stream behind an event exactly as TileGatherer issues it, the two collectives
    the repo uses (all_gather_into_tensor of a tile, all_reduce of the flat gradient buffer) in a world of one.  The multi-rank logic
    around them is covered by the gloo world-2/3 tests (tests/test_dist_cpu.py) and the 8-shard test above."""
    script = tmp_path / "rccl1.py"
    script.write_text('''
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
tile = torch.rand(32768, 3, device="cuda")
out = torch.empty_like(tile)
side = torch.cuda.Stream()
ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream())
with torch.cuda.stream(side):
    side.wait_event(ready)
    dist.all_gather_into_tensor(out, tile)
torch.cuda.current_stream().wait_stream(side)
assert torch.equal(out, tile)
flat = torch.arange(515_000, device="cuda", dtype=torch.float32)
ref = flat.clone()
dist.all_reduce(flat)
assert torch.equal(flat, ref)
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl ok")
''')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "rccl ok" in p.stdout, p.stderr[-2000:]
