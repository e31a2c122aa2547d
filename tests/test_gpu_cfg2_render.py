"""BASELINE cfg2: a generic (non-triplane) NeRF forward render through the OPERATOR API only -- get_encoder('hashgrid') defaults
(D = 3, L = 16, C = 2, T = 2^19, desired resolution 2048), SH(4) directions, small bias-free MLPs, and the reference's inference
loop (nerf_triplane/renderer.py:495-561: march_rays -> network -> composite_rays -> boolean-mask compaction, n_step schedule)
written once and run twice: on the gfx950 operators (torch on the GPU for the Linear layers) and on the CPU checker.  Encoders,
marching and compositing are bit-exact on their own (test_gpu_parity.py); the Linear layers' summation order is the library's, so
whole-frame parity here is the north-star tolerance: RGB / depth within 1e-4, per-ray sample counts equal."""
import numpy as np
import pytest
import torch

from conftest import ellipsoid_bitfield, synthetic_camera
from oracle import oracle as O
from oracle.head import get_rays

pytestmark = pytest.mark.gpu
F32 = np.float32


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda n, k: (torch.rand(n, k, generator=g) * 2 - 1) / k ** 0.5   # torch's default Linear init range
    return dict(s0=mk(64, 32), s1=mk(16, 64), c0=mk(64, 31), c1=mk(3, 64))


class _Cpu:
    def __init__(self, emb, offsets, pls, W):
        self.emb, self.offsets, self.pls = emb, offsets, pls
        self.W = {k: v.numpy() for k, v in W.items()}

    def net(self, xyzs, dirs, bound):
        x01 = (xyzs + F32(bound)) / F32(2 * bound)
        h, _ = O.grid_encode_forward(x01, self.emb, self.offsets, self.pls, 16)
        h = O.linear(O.linear(h, self.W["s0"], relu=True), self.W["s1"])
        sigma = O.unary("exp", np.ascontiguousarray(h[:, 0]))
        d, _ = O.sh_encode_forward(dirs, 4)
        hc = np.ascontiguousarray(np.concatenate([d, h[:, 1:]], 1))
        rgb = O.unary("sigmoid", O.linear(O.linear(hc, self.W["c0"], relu=True), self.W["c1"]))
        return sigma, rgb

    near_far = staticmethod(O.near_far_from_aabb)

    def march(self, n_alive, n_step, alive, t, ro, rd, bound, bits, nears, fars, dt_gamma, max_steps):
        return O.march_rays(n_alive, n_step, alive, t, ro, rd, bound, bits, 1, 128, nears, fars, 128, None, dt_gamma, max_steps)

    def composite(self, n_alive, n_step, alive, t, sig, rgb, dl, ws, dep, img, T):
        O.composite_rays("plain", n_alive, n_step, alive, t, sig, rgb, dl, ws, dep, img, T_thresh=T)

    zeros = staticmethod(lambda *s: np.zeros(s, F32))
    arange = staticmethod(lambda n: np.arange(n, dtype=np.int32))
    compact = staticmethod(lambda a: np.ascontiguousarray(a[a >= 0]))
    count = staticmethod(lambda dl, n_alive, n_step: (dl[: n_alive * n_step, 0] != 0).reshape(n_alive, n_step).sum(1))


class _Gpu:
    def __init__(self, enc, W):
        from lzzx_nerf_amd.encoding import get_encoder
        self.enc = enc
        self.sh = get_encoder("spherical_harmonics")[0]
        self.W = {k: v.cuda() for k, v in W.items()}

    def net(self, xyzs, dirs, bound):
        lin = torch.nn.functional.linear
        h = lin(torch.relu(lin(self.enc(xyzs, bound=bound), self.W["s0"])), self.W["s1"])
        sigma = torch.exp(h[:, 0])
        rgb = torch.sigmoid(lin(torch.relu(lin(torch.cat([self.sh(dirs), h[:, 1:]], -1), self.W["c0"])), self.W["c1"]))
        return sigma, rgb

    def near_far(self, ro, rd, aabb, mn):
        from lzzx_nerf_amd import raymarching as R
        return R.near_far_from_aabb(ro, rd, aabb, mn)

    def march(self, n_alive, n_step, alive, t, ro, rd, bound, bits, nears, fars, dt_gamma, max_steps):
        from lzzx_nerf_amd import raymarching as R
        return R.march_rays(n_alive, n_step, alive, t, ro, rd, bound, bits, 1, 128, nears, fars, 128, False, dt_gamma, max_steps)

    def composite(self, n_alive, n_step, alive, t, sig, rgb, dl, ws, dep, img, T):
        from lzzx_nerf_amd import raymarching as R
        R.composite_rays(n_alive, n_step, alive, t, sig, rgb, dl, ws, dep, img, T)

    zeros = staticmethod(lambda *s: torch.zeros(*s, device="cuda"))
    arange = staticmethod(lambda n: torch.arange(n, dtype=torch.int32, device="cuda"))
    compact = staticmethod(lambda a: a[a >= 0])
    count = staticmethod(lambda dl, n_alive, n_step: (dl[: n_alive * n_step, 0] != 0).reshape(n_alive, n_step).sum(1).cpu().numpy())


def _render(ops, ro, rd, aabb, bits, bound, max_steps, T_thresh=1e-4, dt_gamma=1 / 256):
    """run_cuda, inference branch (renderer.py:495-561), on either backend"""
    N = ro.shape[0]
    nears, fars = ops.near_far(ro, rd, aabb, 0.05)
    ws, dep, img = ops.zeros(N), ops.zeros(N), ops.zeros(N, 3)
    alive = ops.arange(N)
    t = nears.clone() if torch.is_tensor(nears) else nears.copy()
    counts = np.zeros(N, np.int64)
    step = 0
    while step < max_steps:
        n_alive = alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, dl = ops.march(n_alive, n_step, alive, t, ro, rd, bound, bits, nears, fars, dt_gamma, max_steps)
        sig, rgb = ops.net(xyzs, dirs, bound)
        idx = alive.cpu().numpy() if torch.is_tensor(alive) else alive
        np.add.at(counts, idx, ops.count(dl, n_alive, n_step))
        ops.composite(n_alive, n_step, alive, t, sig, rgb, dl, ws, dep, img, T_thresh)
        alive = ops.compact(alive)
        step += n_step
    to_np = lambda a: a.cpu().numpy() if torch.is_tensor(a) else a
    ws, dep, img = to_np(ws), to_np(dep), to_np(img)
    return np.clip(img + (1 - ws)[:, None], 0, 1), dep, ws, counts


@pytest.mark.parametrize("H,max_steps", [(64, 128)])
def test_cfg2_hashgrid_forward_render(H, max_steps):
    from lzzx_nerf_amd.encoding import get_encoder
    enc, out_dim = get_encoder("hashgrid")           # defaults: D3 L16 C2 H16 T19, desired_resolution 2048 (encoding.py:6-8)
    assert out_dim == 32
    enc = enc.cuda()
    rng = np.random.default_rng(11)
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(F32)
    enc.embeddings.data.copy_(torch.from_numpy(emb))
    W = _weights(3)
    W["s1"][0] *= 6.0    # sharper density so that rays terminate early and the n_step schedule / compaction are exercised
    pose, intr = synthetic_camera(H, H)
    ro, rd = get_rays(pose, intr, H, H)
    bits = ellipsoid_bitfield()[0]
    aabb = np.array([-1, -1, -1, 1, 1, 1], F32)
    cpu = _Cpu(emb, enc.offsets.cpu().numpy(), enc.per_level_scale, W)
    img_c, dep_c, ws_c, cnt_c = _render(cpu, ro, rd, aabb, bits, 1.0, max_steps)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    img_g, dep_g, ws_g, cnt_g = _render(_Gpu(enc, W), dev(ro), dev(rd), dev(aabb), dev(bits), 1.0, max_steps)
    assert np.array_equal(cnt_g, cnt_c)                                   # per-ray sample counts
    assert np.abs(img_g - img_c).max() <= 1e-4 and np.abs(dep_g - dep_c).max() <= 1e-4 and np.abs(ws_g - ws_c).max() <= 1e-4
    assert cnt_c.max() > 8 and (ws_c > 0.5).any() and (ws_c < 0.5).any()   # the frame is not trivial


@pytest.mark.parametrize("graph", [False, True])
def test_cfg2_device_resident_loop_equals_the_reference_loop(graph):
    """renderer.NetworkRenderer: the device-resident loop (lz_loop_march -> net on the whole row budget -> lz_loop_composite, no host round
    trip; with graph=True two iterations captured once as a hipGraph and replayed) around the cfg2 network equals the reference's loop on
    the same operators (synthetic.GenericHashgridNeRF.render: per-iteration boolean-mask compaction and host sync) bit for bit -- the
    per-sample network is row-independent (grid encoder, SH, MFMA Linear kernels with a fixed k order) and pixels do not depend on the
    schedule.  Two frames through the same renderer: the second replays the captured graph on new rays."""
    from lzzx_nerf_amd.renderer import NetworkRenderer
    from lzzx_nerf_amd.synthetic import GenericHashgridNeRF
    from lzzx_nerf_amd.utils import frame_rays
    H = 96
    g = GenericHashgridNeRF(torch.device("cuda"))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    bits = dev(ellipsoid_bitfield()[0])
    aabb = dev(np.array([-1, -1, -1, 1, 1, 1], F32))
    r = NetworkRenderer(lambda x, d: g.net(x, d, 1.0), bits, bound=1.0, aabb=aabb, graph=graph)
    for k in (0, 3):
        from lzzx_nerf_amd.synthetic import orbit_pose
        pose, intr = synthetic_camera(H, H)
        if k:
            pose = orbit_pose(k)
        ro, rd = frame_rays(dev(pose), intr, H, H)
        want = g.render(ro, rd, aabb, bits, max_steps=128)
        got = r.render(ro, rd, max_steps=128, count_samples=True)
        assert torch.equal(got["image"], want[0]), k
        assert torch.equal(got["depth"], want[1]) and torch.equal(got["weights_sum"], want[2])
        assert int(got["state"][3]) == 1 and int(got["ray_counts"].sum()) == int(got["state"][5]) > 10000
        assert float(want[2].max()) > 0.5 and float(want[2].min()) < 0.5


def test_network_renderer_small_batches_and_recapture():
    """NetworkRenderer edge cases: a handful of rays (one workgroup of the loop kernels), a ray count change between frames (buffers and the
    captured graph are rebuilt), rays that all miss the box (no sample: background), and a budget of one row per ray (the reference's)."""
    from lzzx_nerf_amd.renderer import NetworkRenderer
    from lzzx_nerf_amd.synthetic import GenericHashgridNeRF
    from lzzx_nerf_amd.utils import frame_rays
    g = GenericHashgridNeRF(torch.device("cuda"))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    bits = dev(ellipsoid_bitfield()[0])
    aabb = dev(np.array([-1, -1, -1, 1, 1, 1], F32))
    pose, intr = synthetic_camera(40, 40)
    ro, rd = frame_rays(dev(pose), intr, 40, 40)
    r = NetworkRenderer(lambda x, d: g.net(x, d, 1.0), bits, bound=1.0, aabb=aabb, graph=True)
    for sel in (slice(0, 1600), slice(780, 787), slice(0, 1600, 3)):
        want = g.render(ro[sel].contiguous(), rd[sel].contiguous(), aabb, bits, max_steps=64)
        got = r.render(ro[sel], rd[sel], max_steps=64)
        assert torch.equal(got["image"], want[0]) and torch.equal(got["weights_sum"], want[2]), sel
    up = torch.zeros_like(rd)
    up[:, 1] = 1.0
    up[:, 0] = up[:, 2] = 1e-3
    out = r.render(ro + torch.tensor([0.0, 5.0, 0.0], device="cuda"), up, max_steps=64)
    assert float(out["image"].min()) == 1.0 and int(out["state"][5]) == 0
    r1 = NetworkRenderer(lambda x, d: g.net(x, d, 1.0), bits, bound=1.0, aabb=aabb, budget_factor=1, n_step_cap=8, graph=False)
    want = g.render(ro, rd, aabb, bits, max_steps=64)
    got = r1.render(ro, rd, max_steps=64, count_samples=True)
    assert torch.equal(got["image"], want[0]) and int(got["state"][72]) == want[3]      # the reference's schedule: the same rows, too


def test_network_renderer_graph_follows_a_swapped_bitfield_and_schedule():
    """the captured hipGraph bakes in the bitfield / aabb / buffer addresses and the schedule; replacing `renderer.bitfield` (an
    occupancy update) or changing budget_factor / n_step_cap between frames must re-capture, not replay launches on stale pointers."""
    from lzzx_nerf_amd.renderer import NetworkRenderer
    from lzzx_nerf_amd.synthetic import GenericHashgridNeRF
    from lzzx_nerf_amd.utils import frame_rays
    g = GenericHashgridNeRF(torch.device("cuda"))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    bits_a = dev(ellipsoid_bitfield()[0])
    bits_b = dev(ellipsoid_bitfield(semi=(0.2, 0.25, 0.5))[0])
    aabb = dev(np.array([-1, -1, -1, 1, 1, 1], F32))
    pose, intr = synthetic_camera(64, 64)
    ro, rd = frame_rays(dev(pose), intr, 64, 64)
    r = NetworkRenderer(lambda x, d: g.net(x, d, 1.0), bits_a, bound=1.0, aabb=aabb, graph=True)
    a = r.render(ro, rd, max_steps=64)["image"].clone()
    assert torch.equal(a, g.render(ro, rd, aabb, bits_a, max_steps=64)[0])
    r.bitfield = bits_b                                  # a new tensor: the old graph points at bits_a
    b = r.render(ro, rd, max_steps=64)["image"].clone()
    assert torch.equal(b, g.render(ro, rd, aabb, bits_b, max_steps=64)[0]) and not torch.equal(a, b)
    r.budget_factor, r.n_step_cap = 1, 8                 # another row budget: new buffers, new graph
    c = r.render(ro, rd, max_steps=64, count_samples=True)
    want = g.render(ro, rd, aabb, bits_b, max_steps=64)
    assert torch.equal(c["image"], want[0]) and int(c["state"][72]) == want[3]
