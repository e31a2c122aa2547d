"""Randomised differential test of the headline path: TriplaneRenderer(mode="fused", cap="reference") -- one persistent kernel, the
frame-wide cap C_eff, any steps_per_pass -- against the multi-launch loop under the reference's schedule (budget_factor, n_step_cap) =
(1, 8) (renderer.py:503-548), which the parity tests hold to the CPU checker and to the reference-run fixtures.  Seeded draws over frame
size, occupancy, max_steps (1 .. 48: the cap binds in most cases), T_thresh, dt_gamma, launch shape, precision, perturbed starts, a
background image, the march confined to the occupied bounds; every output and the per-ray marched counts must be identical."""
import os

import numpy as np
import pytest
import torch

from conftest import ellipsoid_bitfield, synthetic_camera

pytestmark = pytest.mark.gpu
KEYS = ("image", "image_raw", "weights_sum", "depth", "amb_aud_sum", "amb_eye_sum", "uncertainty_sum", "nears", "fars", "ray_counts")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(rng, kind):
    n = 128 ** 3
    if kind == "ones":
        return np.full(n // 8, 255, np.uint8)
    if kind == "empty":
        return np.zeros(n // 8, np.uint8)
    if kind == "ellipsoid":
        return ellipsoid_bitfield()[0]
    if kind == "dust":          # every cell occupied with probability p: rays alternate between samples and one-cell skips
        p = rng.choice([0.02, 0.3, 0.7])
        return np.packbits(rng.random(n) < p, bitorder="little")
    # slabs: a few occupied byte runs in Morton order = spatially compact blocks with empty space between them
    bits = np.zeros(n // 8, np.uint8)
    for _ in range(int(rng.integers(2, 9))):
        a = int(rng.integers(0, n // 8 - 4096))
        bits[a:a + int(rng.integers(64, 4096))] = 255
    return bits


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    c = dict(H=int(rng.integers(9, 90)), W=int(rng.integers(9, 90)), scene=str(rng.choice(["ones", "ellipsoid", "dust", "slabs", "slabs", "empty"], p=[.2, .25, .2, .15, .15, .05])),
             max_steps=int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 31, 32, 48])), T_thresh=float(rng.choice([1e-4, 1e-4, 0.3, 0.8])),
             dt_gamma=float(rng.choice([0.0, 1 / 256, 1 / 256, 1 / 64])), S=int(rng.choice([0, 1, 2, 4, 8, 16])),
             precision=str(rng.choice(["f32", "f32", "f16"])), noise=bool(rng.random() < 0.3), bg=bool(rng.random() < 0.3), occ=bool(rng.random() < 0.5))
    return rng, c


@pytest.mark.parametrize("seed", range(int(os.environ.get("LZ_RANDOM_FRAMES", "48"))))     # LZ_RANDOM_FRAMES=600: the soak run of round 4
def test_fused_reference_cap_equals_the_loop_on_random_frames(params, golden, seed):
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    from lzzx_nerf_amd.utils import frame_rays
    rng, c = _case(seed)
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0, precision=c["precision"])
    bits = dev(_bits(rng, c["scene"]))
    pose, intr = synthetic_camera(c["H"], c["W"])
    ro, rd = frame_rays(dev(pose), intr, c["H"], c["W"])
    N = ro.shape[0]
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    kw = dict(max_steps=c["max_steps"], T_thresh=c["T_thresh"], dt_gamma=c["dt_gamma"], count_samples=True)
    if c["noise"]:
        kw["noises"] = dev(rng.uniform(0, 1, N).astype(np.float32))
    if c["bg"]:
        kw["bg_color"] = dev(rng.uniform(0, 1, (N, 3)).astype(np.float32))
    fr = TriplaneRenderer(head, bits, bound=1.0, mode="fused", cap="reference")
    fr.steps_per_pass = c["S"]
    fr.clip_to_occupancy = c["occ"]
    fused = {k: v.clone() for k, v in fr.render(ro, rd, *cond, **kw).items()}
    loop = TriplaneRenderer(head, bits, bound=1.0, budget_factor=1, n_step_cap=8).render(ro, rd, *cond, **kw)
    for k in KEYS:
        assert torch.equal(fused[k], loop[k]), (c, k, int((fused[k] != loop[k]).sum()))
    assert int(fused["state"][5]) == int(loop["state"][5]) == int(fused["ray_counts"].sum()), c


@pytest.mark.parametrize("seed", range(int(os.environ.get("LZ_RANDOM_TILED_FRAMES", "16"))))
def test_tiles_of_random_frames_equal_the_unsharded_reference_loop(params, golden, seed):
    """The same draws rendered as 2 .. 8 tiles of ONE frame (dist.ShardedFrame; every tile by its own fused renderer with its own launch
    shape, the max_steps + 1 histogram words of the first phases summed as the all-reduce does): the assembled image, depth and marched
    counts equal the unsharded loop under the reference schedule -- n_alive / N of renderer.py:513 are frame-wide quantities."""
    from lzzx_nerf_amd import dist as D
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    from lzzx_nerf_amd.utils import frame_rays
    rng, c = _case(5000 + seed)
    world = int(rng.integers(2, 9))
    tiles = str(rng.choice(["interleaved", "contiguous"]))
    H, W = max(c["H"], 2 * world), c["W"]
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0, precision=c["precision"])
    bits = dev(_bits(rng, c["scene"]))
    pose, intr = synthetic_camera(H, W)
    ro, rd = frame_rays(dev(pose), intr, H, W)
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    kw = dict(max_steps=c["max_steps"], T_thresh=c["T_thresh"], dt_gamma=c["dt_gamma"], count_samples=True)
    noise = dev(rng.uniform(0, 1, H * W).astype(np.float32)) if c["noise"] else None
    ref = TriplaneRenderer(head, bits, bound=1.0).render(ro, rd, *cond, noises=noise, **kw)
    ref = {k: v.clone() for k, v in ref.items()}
    rs, ctxs = [], []
    for g in range(world):
        sf = D.ShardedFrame(H, W, g, world, tiles, device="cuda")
        sf.gatherer = None
        r = sf.configure(TriplaneRenderer(head, bits, bound=1.0, mode="fused"))
        r.steps_per_pass = int(rng.choice([0, 0, 1, 2, 4, 8, 16]))
        r.clip_to_occupancy = c["occ"]
        px = sf.pixels
        tro, trd = sf.rays(dev(pose), intr)          # ray generation for the tile's pixels (an EMPTY tile when the frame has fewer row blocks than ranks)
        assert torch.equal(tro, ro[px]) and torch.equal(trd, rd[px])
        ctxs.append(r.fused_begin(tro, trd, *cond, noises=None if noise is None else noise[px].contiguous(), **kw))
        rs.append(r)
    total = torch.stack([x["hist"] for x in ctxs]).sum(0)
    assert int(total.sum()) == H * W, (c, world, tiles)
    outs = []
    for r, x in zip(rs, ctxs):
        x["hist"].copy_(total)
        outs.append({k: v.clone() for k, v in r.fused_finish(x).items()})
    for k in ("image", "depth", "weights_sum", "ray_counts"):
        cat = torch.cat([o[k] for o in outs])
        got = D.assemble_frame(cat if cat.dim() > 1 else cat[:, None], H, W, world, tiles)
        got = got if cat.dim() > 1 else got[:, 0]
        assert torch.equal(got, ref[k]), (c, world, tiles, k, int((got != ref[k]).sum()))


@pytest.mark.parametrize("seed", range(int(os.environ.get("LZ_RANDOM_LOOP_FRAMES", "12"))))
def test_loop_mode_on_random_schedules_equals_the_checker(params, golden, seed):
    """mode="loop" (the device-resident form of renderer.py:503-548) under a random (budget_factor, n_step_cap) against the CPU checker's
    loop under the same schedule: pixels, depth, sums, iteration count and per-ray marched counts bit for bit -- and the fused frame with
    cap = "per_ray" and steps_per_pass = S against the checker under (S, S)"""
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    from lzzx_nerf_amd.utils import frame_rays
    from oracle.head import TriplaneSpec
    from oracle.render import render_inference
    rng, c = _case(9000 + seed)
    H, W = min(c["H"], 40), min(c["W"], 40)                  # the checker's head is a CPU loop: keep the frames small
    bf, cap_n = int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 4, 8]))
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0)
    bits_np = _bits(rng, c["scene"])
    bits = dev(bits_np)
    pose, intr = synthetic_camera(H, W)
    ro, rd = frame_rays(dev(pose), intr, H, W)
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    kw = dict(max_steps=c["max_steps"], T_thresh=c["T_thresh"], dt_gamma=c["dt_gamma"])
    noise = rng.uniform(0, 1, H * W).astype(np.float32) if c["noise"] else None
    st = {}
    ref = render_inference(TriplaneSpec(1.0), params, ro.cpu().numpy(), rd.cpu().numpy(), bits_np, golden["net_enc_a"], golden["net_ind"],
                           golden["net_eye"], stats=st, budget_factor=bf, n_step_cap=cap_n, noises=noise, **kw)
    loop = TriplaneRenderer(head, bits, bound=1.0, budget_factor=bf, n_step_cap=cap_n).render(
        ro, rd, *cond, count_samples=True, noises=None if noise is None else dev(noise), **kw)
    for k in ("image", "depth", "weights_sum", "amb_aud_sum", "amb_eye_sum", "uncertainty_sum"):
        assert np.array_equal(loop[k].cpu().numpy(), ref[k]), (c, bf, cap_n, k)
    assert np.array_equal(loop["ray_counts"].cpu().numpy().astype(np.int64), st["samples_per_ray"]), (c, bf, cap_n)
    assert int(loop["state"][6]) == len(st["schedule"]), (c, bf, cap_n)
    S = int(rng.choice([1, 2, 4, 8, 16]))
    st2 = {}
    ref2 = render_inference(TriplaneSpec(1.0), params, ro.cpu().numpy(), rd.cpu().numpy(), bits_np, golden["net_enc_a"], golden["net_ind"],
                            golden["net_eye"], stats=st2, budget_factor=S, n_step_cap=S, noises=noise, **kw)
    fr = TriplaneRenderer(head, bits, bound=1.0, mode="fused", cap="per_ray")
    fr.steps_per_pass = S
    fused = fr.render(ro, rd, *cond, count_samples=True, noises=None if noise is None else dev(noise), **kw)
    for k in ("image", "depth", "weights_sum"):
        assert np.array_equal(fused[k].cpu().numpy(), ref2[k]), (c, S, k)
    assert np.array_equal(fused["ray_counts"].cpu().numpy().astype(np.int64), st2["samples_per_ray"]), (c, S)
