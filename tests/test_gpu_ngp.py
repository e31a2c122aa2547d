"""BASELINE cfg2 on the fused path (lzzx_nerf_amd/ngp.py, csrc/lz_ngp.hip): the tiled level-major gather, the one-kernel hash-grid NeRF
head and the 4-launch device-resident loop, against the CPU checker (oracle/ngp.py: the same operators with the kernel's summation
order spelled out -- bit for bit) and against the operator-API network the fused head replaces (synthetic.GenericHashgridNeRF on the
lz_linear kernels: another summation order, so north_star's 1e-4 and equal per-ray counts)."""
import numpy as np
import pytest
import torch

from conftest import ellipsoid_bitfield, synthetic_camera
from oracle import ngp as ONGP
from oracle import oracle as O

pytestmark = pytest.mark.gpu
F32 = np.float32


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make(seed=3, half=False):
    from lzzx_nerf_amd.ngp import FusedHashgridNeRF
    from lzzx_nerf_amd.synthetic import GenericHashgridNeRF
    g = GenericHashgridNeRF(torch.device("cuda"), seed=seed, half_tables=half)
    fused = FusedHashgridNeRF(g.enc, g.sigma_net, g.color_net, half_tables=half)
    W = dict(s0=g.sigma_net.net[0].weight, s1=g.sigma_net.net[1].weight, c0=g.color_net.net[0].weight, c1=g.color_net.net[1].weight)
    W = {k: v.detach().cpu().numpy() for k, v in W.items()}
    emb = g.enc.embeddings.detach().cpu().numpy()
    if half:
        emb = emb.astype(np.float16)
    cpu_net = ONGP.network(W, emb, g.enc.offsets.cpu().numpy(), g.enc.per_level_scale)
    return g, fused, W, cpu_net


@pytest.mark.parametrize("M", [5, 1000, 70001])
@pytest.mark.parametrize("half", [False, True])
def test_tiled_gather_equals_the_row_major_encoder(M, half):
    """lz_grid_encode_forward_tiled leaves [tile][level][sample][C]; re-ordered it is the module's [B, L*C] output bit for bit (f32 and half
    tables), the (x + bound) / (2 bound) mapping included; rows behind a device-side count are not touched"""
    g, fused, _, _ = make(half=half)
    rng = np.random.default_rng(M)
    # bound 2 with points up to 2.3: some fall outside -> zero features (gridencoder.cu:98-122).  (A power of two: torch's GPU division by a
    # python scalar multiplies by the rounded reciprocal, the kernel and the checker divide -- the two agree when 1 / (2 bound) is exact.)
    x = dev(rng.uniform(-2.3, 2.3, (M, 3)).astype(F32))
    want = g.enc(x, bound=2.0) if not half else None
    if half:
        with torch.autocast("cuda", dtype=torch.float16):
            want = g.enc(x, bound=2.0)
    feats = torch.full((M, 32), 7.0, dtype=fused.table.dtype, device="cuda")
    fused.encode_tiled(x, feats, 2.0)
    flat = feats.reshape(-1)
    rows = []
    for b0 in range(0, M, 256):
        n = min(256, M - b0)
        rows.append(flat[b0 * 32: (b0 + n) * 32].reshape(16, n, 2).permute(1, 0, 2).reshape(n, 32))
    got = torch.cat(rows)
    assert got.dtype == want.dtype and torch.equal(got, want)
    if M > 600:
        cnt = torch.tensor([300], dtype=torch.int32, device="cuda")
        f2 = torch.full((M, 32), 7.0, dtype=fused.table.dtype, device="cuda")
        fused.encode_tiled(x, f2, 2.0, count_ptr=cnt.data_ptr())
        assert torch.equal(f2.reshape(-1)[: 512 * 32], flat[: 512 * 32]) and bool((f2.reshape(-1)[512 * 32:] == 7.0).all())


@pytest.mark.parametrize("M", [3, 16, 1000, 66000])
@pytest.mark.parametrize("half", [False, True])
def test_fused_head_bit_exact_vs_checker(M, half):
    g, fused, W, cpu_net = make(half=half)
    rng = np.random.default_rng(M + 1)
    x = rng.uniform(-1, 1, (M, 3)).astype(F32)
    d = rng.normal(size=(M, 3)).astype(F32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sig, rgb = fused.forward(dev(x), dev(d), 1.0)
    sig_c, rgb_c = cpu_net(x, d, 1.0)
    assert np.array_equal(sig.cpu().numpy(), sig_c) and np.array_equal(rgb.cpu().numpy(), rgb_c)
    # the operator-API network the kernel replaces: same values up to the summation order of its Linear kernels
    sig_o, rgb_o = g.net(dev(x), dev(d), 1.0)
    assert float(((sig - sig_o).abs() / (1 + sig_o.abs())).max()) < 1e-5 and float((rgb - rgb_o).abs().max()) < 1e-5
    # row-major features (layout 0) through the same kernel
    feats = g.enc(dev(x), bound=1.0).float().contiguous() if not half else None
    if feats is not None:
        from lzzx_nerf_amd._util import call, ptr, stream
        s0, r0 = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
        call("lz_ngp_head_forward", ptr(fused.packed), ptr(feats), 0, ptr(dev(d)), M, None, ptr(s0), ptr(r0), stream())
        assert torch.equal(s0, sig) and torch.equal(r0, rgb)


def test_pack_weights_rejects_other_architectures():
    from lzzx_nerf_amd.ngp import pack_weights
    w = lambda n, k: torch.zeros(n, k, device="cuda")
    with pytest.raises(RuntimeError, match="32-64-16"):
        pack_weights(w(64, 32), w(16, 64), w(64, 32), w(3, 64))


@pytest.mark.parametrize("schedule", [(1, 8), (8, 8), (4, 4)])
@pytest.mark.parametrize("H,max_steps,half", [(64, 128, False), (96, 16, False), (64, 64, True)])
def test_hashgrid_renderer_equals_the_reference_loop(H, max_steps, half, schedule):
    """HashgridRenderer against the reference's loop (renderer.py:495-561) written on the CPU checker with the fused head's arithmetic
    (tests/test_gpu_cfg2_render._render): under the reference's schedule (1, 8) image, depth, weights and per-ray counts bit for bit, the
    cap at max_steps = 16 included; other schedules keep the pixels of every ray that ends before the cap.  And against the operator-API
    network under the reference's loop with a host sync per iteration (synthetic.GenericHashgridNeRF.render): 1e-4, same weights."""
    from test_gpu_cfg2_render import _Cpu, _render
    from lzzx_nerf_amd.ngp import HashgridRenderer
    from lzzx_nerf_amd.utils import frame_rays
    g, fused, W, cpu_net = make(half=half)
    bits = ellipsoid_bitfield()[0]
    aabb = np.array([-1, -1, -1, 1, 1, 1], F32)
    pose, intr = synthetic_camera(H, H)
    ro, rd = frame_rays(dev(pose), intr, H, H)
    r = HashgridRenderer(fused, dev(bits), bound=1.0, aabb=dev(aabb), budget_factor=schedule[0], n_step_cap=schedule[1])
    got = {k: v.clone() for k, v in r.render(ro, rd, max_steps=max_steps, count_samples=True).items()}

    class Cpu(_Cpu):
        def __init__(self):
            pass

        def net(self, xyzs, dirs, bound):
            return cpu_net(xyzs, dirs, bound)
    img_c, dep_c, ws_c, cnt_c = _render(Cpu(), ro.cpu().numpy(), rd.cpu().numpy(), aabb, bits, 1.0, max_steps)
    if schedule == (1, 8):
        assert np.array_equal(got["image"].cpu().numpy(), img_c.astype(F32))
        assert np.array_equal(got["depth"].cpu().numpy(), dep_c) and np.array_equal(got["weights_sum"].cpu().numpy(), ws_c)
        assert np.array_equal(got["ray_counts"].cpu().numpy().astype(np.int64), cnt_c)
        if max_steps == 16:
            assert cnt_c.max() > 16                                   # the cap binds: C_eff = sum of n_step
    elif max_steps > 16:
        assert np.array_equal(got["image"].cpu().numpy(), img_c.astype(F32))     # pixels do not depend on the schedule
    assert int(got["state"][3]) == 1 and int(got["state"][5]) == int(got["ray_counts"].sum())
    if schedule == (1, 8) and not half:
        want = g.render(ro, rd, dev(aabb), dev(bits), max_steps=max_steps)
        assert float((got["image"] - want[0]).abs().max()) <= 1e-4 and float((got["weights_sum"] - want[2]).abs().max()) <= 1e-4
    assert float(got["weights_sum"].max()) > 0.5
