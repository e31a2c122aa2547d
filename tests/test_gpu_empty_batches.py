"""A batch of NO rays is a legal input of every renderer: a rank whose tile of a small frame is empty (dist.tile_rows deals 8-row stripes:
a 16-row frame leaves ranks 2 .. 7 of 8 without a row) still calls the renderer and still takes part in the frame's collectives.  Found
by tests/test_gpu_fused_random.py (round 4): the C entries rejected the empty tensors' null pointers before looking at N."""
import numpy as np
import pytest
import torch

from conftest import ellipsoid_bitfield, synthetic_camera

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("mode,cap", [("loop", "reference"), ("fused", "reference"), ("fused", "per_ray")])
def test_triplane_renderer_on_no_rays(params, golden, mode, cap):
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0)
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    e = torch.empty(0, 3, device="cuda")
    r = TriplaneRenderer(head, dev(ellipsoid_bitfield()[0]), bound=1.0, mode=mode, cap=cap)
    o = r.render(e, e, *cond, max_steps=16, count_samples=True, rgb24=True)
    torch.cuda.synchronize()
    assert o["image"].shape == (0, 3) and o["depth"].shape == (0,) and o["ray_counts"].shape == (0,) and o["image_rgb24"].shape == (0, 3)
    assert int(o["state"][5]) == 0


def test_an_empty_tile_takes_part_in_the_cap_exchange(params, golden):
    """fused_begin on no rays leaves a ZERO histogram for the frame's all-reduce (not the previous frame's), fused_finish returns empty
    outputs; ray generation for an empty pixel selection is a no-op too"""
    from lzzx_nerf_amd import dist as D
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0)
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    H, W, world = 16, 24, 4
    assert [len(D.tile_rows(H, g, world, "interleaved")) for g in range(world)] == [8, 8, 0, 0]
    pose, intr = synthetic_camera(H, W)
    bits = dev(ellipsoid_bitfield()[0])
    sf = D.ShardedFrame(H, W, 3, world, "interleaved", device="cuda")
    sf.gatherer = None
    r = sf.configure(TriplaneRenderer(head, bits, bound=1.0, mode="fused"))
    # the renderer's buffers carry an older frame's histogram: render a non-empty batch of the same size class first
    full = D.ShardedFrame(H, W, 0, world, "interleaved", device="cuda")
    ro, rd = full.rays(dev(pose), intr)
    c0 = r.fused_begin(ro, rd, *cond, max_steps=16, count_samples=True)
    assert int(c0["hist"].sum()) == ro.shape[0]
    r.fused_finish(c0)
    ro, rd = sf.rays(dev(pose), intr)
    assert ro.shape == (0, 3) and rd.shape == (0, 3)
    ctx = r.fused_begin(ro, rd, *cond, max_steps=16, count_samples=True)
    assert ctx["deferred"] and int(ctx["hist"].abs().sum()) == 0
    o = r.fused_finish(ctx)
    torch.cuda.synchronize()
    assert o["image"].shape == (0, 3) and o["ray_counts"].shape == (0,)


def test_hashgrid_renderers_on_no_rays():
    from lzzx_nerf_amd.ngp import FusedHashgridNeRF, HashgridRenderer
    from lzzx_nerf_amd.renderer import NetworkRenderer
    from lzzx_nerf_amd.synthetic import GenericHashgridNeRF
    g = GenericHashgridNeRF(torch.device("cuda"), seed=3)
    fused = FusedHashgridNeRF(g.enc, g.sigma_net, g.color_net)
    bits = dev(ellipsoid_bitfield()[0])
    e = torch.empty(0, 3, device="cuda")
    for r in (HashgridRenderer(fused, bits, bound=1.0), NetworkRenderer(lambda x, d: g(x, d), bits, bound=1.0)):
        o = r.render(e, e, max_steps=32, count_samples=True)
        torch.cuda.synchronize()
        assert o["image"].shape == (0, 3) and o["ray_counts"].shape == (0,)
