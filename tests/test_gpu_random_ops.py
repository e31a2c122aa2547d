"""Randomised differential tests of the operator kernels against the CPU checker: seeded draws over shapes and parameters the fixed case
lists of test_gpu_parity.py do not visit (batch sizes around the kernels' tile edges, odd level counts, tiny tables, both grid types,
align_corners, overflowing sample buffers, random chunk shapes of the inference step).  Same bars as the fixed cases: integer outputs
and every value both sides compute with the same operation sequence bit for bit; atomic accumulations to the reassociation budget."""
import os

import numpy as np
import pytest
import torch

from conftest import ellipsoid_bitfield
from oracle import oracle as O

pytestmark = pytest.mark.gpu
NCASE = int(os.environ.get("LZ_RANDOM_OPS", "24"))
BATCHES = [1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 4099, 32767, 32768, 32769, 50021]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _grid_case(seed):
    rng = np.random.default_rng(7000 + seed)
    D = int(rng.choice([1, 2, 2, 3, 3, 4]))
    c = dict(D=D, L=int(rng.integers(1, 17)), C=int(rng.choice([1, 2, 2, 4, 8])), H=int(rng.choice([2, 3, 4, 8, 16, 17, 32, 64])),
             T=int(rng.integers(6, 20)), res=(None if rng.random() < 0.25 else int(rng.choice([16, 64, 200, 512, 2048]))),
             gt=str(rng.choice(["hash", "tiled"])), ac=bool(rng.random() < 0.3), B=int(rng.choice(BATCHES)))
    if c["res"] is not None and c["res"] < c["H"]:
        c["res"] = c["H"] * 2
    if c["L"] == 1:
        c["res"] = None          # grid.py:95-96 divides by num_levels - 1
    return rng, c


@pytest.mark.parametrize("seed", range(NCASE))
def test_grid_encoder_on_random_shapes(seed):
    from lzzx_nerf_amd._util import call, ptr, stream
    from lzzx_nerf_amd.gridencoder import GridEncoder, grid_encode
    rng, c = _grid_case(seed)
    D, L, C, H, B = c["D"], c["L"], c["C"], c["H"], c["B"]
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=c["T"], desired_resolution=c["res"],
                      gridtype=c["gt"], align_corners=c["ac"]).cuda()
    gid = 0 if c["gt"] == "hash" else 1
    off = host(enc.offsets)
    assert np.array_equal(off, O.grid_offsets(D, L, enc.per_level_scale, H, c["T"], c["ac"])), c
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    edge = rng.integers(0, B, min(B, 6))
    x[edge[: len(edge) // 2]] = rng.choice([0.0, 1.0, 0.5], (len(edge) // 2, D)).astype(np.float32)
    if B > 3:
        x[edge[-1], 0] = np.float32(1.0000001)
    xt, et = dev(x), dev(emb)
    out_o, dd_o = O.grid_encode_forward(x, emb, off, enc.per_level_scale, H, True, gid, c["ac"])
    for calc in (True, False):      # with dy_dx: the level-major reference kernel; without: the wrapper's own choice of hot kernel
        out = grid_encode(xt, et, enc.offsets, enc.per_level_scale, H, calc, gid, c["ac"])
        assert np.array_equal(host(out), out_o), (c, calc, int((host(out) != out_o).sum()))
    S = float(np.float32(np.log2(enc.per_level_scale)))
    idx = torch.empty(L, B, 1 << D, dtype=torch.int32, device="cuda")
    call("lz_grid_corner_indices", ptr(xt), ptr(enc.offsets), ptr(idx), B, D, C, L, S, H, gid, int(c["ac"]), stream())
    assert np.array_equal(host(idx), O.grid_corner_indices(x, off, C, enc.per_level_scale, H, gid, c["ac"])), c
    dd = torch.empty(B, L * D * C, device="cuda")
    out_lm = torch.empty(L, B, C, device="cuda")
    call("lz_grid_encode_forward", ptr(xt), ptr(et), ptr(enc.offsets), ptr(out_lm), B, D, C, L, S, H, ptr(dd), gid, int(c["ac"]), 0, 0, stream())
    assert np.array_equal(host(dd), dd_o), c
    # backward: table gradient (atomics: reassociation budget) and input gradient (sequential: bits), every gradient layout
    g = rng.normal(size=(B, L * C)).astype(np.float32)
    ge, gi = O.grid_encode_backward(g, x, tuple(emb.shape), off, enc.per_level_scale, H, dd_o, gid, c["ac"])
    scale = max(float(np.abs(ge).max()), 1e-30)
    gt_ = dev(g)
    for layout in (0, 1, 2, 3):
        gin = gt_ if layout in (1, 2) else gt_.view(B, L, C).permute(1, 0, 2).contiguous()
        gemb = torch.zeros_like(et)
        ginp = torch.zeros(B, D, device="cuda")
        call("lz_grid_encode_backward", ptr(gin), ptr(xt), ptr(et), ptr(enc.offsets), ptr(gemb), B, D, C, L, S, H, ptr(dd), ptr(ginp), gid,
             int(c["ac"]), 0, layout, stream())
        # (the checker adds in f32 in sample order: its own rounding walk grows with the terms per entry -- small tables under 50 000 samples)
        assert np.max(np.abs(host(gemb) - ge)) <= (2e-5 if B <= 5000 else 1e-4) * scale, (c, layout)
        assert np.array_equal(host(ginp), gi), (c, layout)


@pytest.mark.parametrize("seed", range(NCASE))
def test_sh_and_freq_encoders_on_random_shapes(seed):
    from lzzx_nerf_amd.freqencoder import freq_encode
    from lzzx_nerf_amd.shencoder import sh_encode
    rng = np.random.default_rng(8000 + seed)
    B = int(rng.choice(BATCHES))
    deg = int(rng.integers(1, 9))
    d = rng.normal(size=(B, 3)).astype(np.float32)
    if rng.random() < 0.5:
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    dt = dev(d).requires_grad_(True)
    out = sh_encode(dt, deg, True)
    out_o, dd_o = O.sh_encode_forward(d, deg, True)
    assert np.array_equal(host(out), out_o), (B, deg)
    g = rng.normal(size=out_o.shape).astype(np.float32)
    out.backward(dev(g))
    assert np.array_equal(host(dt.grad), O.sh_encode_backward(g, dd_o, deg)), (B, deg)
    Din, fdeg = int(rng.integers(1, 7)), int(rng.integers(1, 11))
    x = rng.uniform(-2, 2, (B, Din)).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    fo = freq_encode(xt, fdeg, Din + Din * 2 * fdeg)
    fo_o = O.freq_encode_forward(x, fdeg)
    assert np.array_equal(host(fo), fo_o), (B, Din, fdeg)
    gf = rng.normal(size=fo_o.shape).astype(np.float32)
    fo.backward(dev(gf))
    assert np.array_equal(host(xt.grad), O.freq_encode_backward(gf, fo_o, Din, fdeg)), (B, Din, fdeg)


def _rays(rng, N, bound):
    """rays from points around the box towards points inside it (most hit, some graze or miss)"""
    o = rng.normal(size=(N, 3)).astype(np.float32)
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(1.2, 3.5, (N, 1)) * bound).astype(np.float32)
    tgt = rng.uniform(-1.1, 1.1, (N, 3)).astype(np.float32) * np.float32(bound)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    if N > 4:
        d[0] = [0, 0, 1]; o[0] = [0.1, 0.1, -2.5 * bound]            # axis-parallel: infinite reciprocals in the slab test
        d[1] = [1, 0, 0]; o[1] = [-3 * bound, 5 * bound, 0]          # ... and a miss
    return o, d


def _bitfield(rng, cascade):
    n = cascade * 128 ** 3
    kind = rng.choice(["ones", "ellipsoid", "dust", "slabs"])
    if kind == "ones":
        return np.full(n // 8, 255, np.uint8)
    if kind == "ellipsoid":
        return np.concatenate([ellipsoid_bitfield()[0]] * cascade)
    if kind == "dust":
        return np.packbits(rng.random(n) < rng.choice([0.05, 0.4]), bitorder="little")
    bits = np.zeros(n // 8, np.uint8)
    for _ in range(int(rng.integers(3, 12))):
        a = int(rng.integers(0, n // 8 - 8192))
        bits[a:a + int(rng.integers(64, 8192))] = 255
    return bits


@pytest.mark.parametrize("seed", range(NCASE))
def test_march_rays_train_on_random_rays(seed):
    """march_rays_train + its backward (raymarching.py:186-280): rays, samples and counters bit for bit, with perturbation, for buffers
    sized by the counter (force_all_rays), by a mean_count that fits and by one that OVERFLOWS (rays dropped, raymarching.cu:457)"""
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd._util import call, ptr, stream
    rng = np.random.default_rng(9000 + seed)
    cascade = int(rng.choice([1, 1, 2, 3]))
    bound = float(rng.choice([1.0, 1.0, 1.5, 2.0, 4.0])) if cascade > 1 else float(rng.choice([0.7, 1.0]))
    N = int(rng.choice([1, 5, 63, 64, 65, 255, 256, 257, 1000, 2311]))
    ro, rd = _rays(rng, N, bound)
    aabb = (np.array([-1, -0.6, -1, 1, 0.8, 1], np.float32) * np.float32(bound)).astype(np.float32)
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.05)
    bits = _bitfield(rng, cascade)
    max_steps = int(rng.choice([4, 16, 48, 100]))
    dt_gamma = float(rng.choice([0.0, 1 / 256, 1 / 32]))
    noises = rng.uniform(0, 1, N).astype(np.float32)
    n_g, f_g = R.near_far_from_aabb(dev(ro), dev(rd), dev(aabb), 0.05)
    assert np.array_equal(host(n_g), nears) and np.array_equal(host(f_g), fars)
    # total sample count first (force_all_rays), then the three sizings through the C ABI with the recorded noise
    c0 = np.zeros(2, np.int32)
    x_all, _, _, _ = O.march_rays_train(ro, rd, bound, bits, cascade, 128, nears, fars, c0, -1, noises, 128, True, dt_gamma, max_steps)
    total = int(c0[0])
    keep = [dev(a) for a in (ro, rd, bits, nears, fars, noises)]
    for M in sorted({total + 128 - total % 128, max(total // 2, 1), total + 7}):
        xo, do, lo, ro_ = O.march_rays_train(ro, rd, bound, bits, cascade, 128, nears, fars, np.zeros(2, np.int32), M, noises, -1, False, dt_gamma, max_steps)
        assert xo.shape[0] == M
        # NaN-filled buffers: the entry zeroes every row it does not write itself (tail, dropped rays) -- the caller no longer has to
        xt, dt_, lt = (torch.full(sh, float("nan"), device="cuda") for sh in ((M, 3), (M, 3), (M, 2)))
        rt = torch.empty(N, 3, dtype=torch.int32, device="cuda")
        ctr = torch.zeros(2, dtype=torch.int32, device="cuda")
        ws = torch.empty(N + 2, dtype=torch.int32, device="cuda")
        call("lz_march_rays_train", ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), bound, dt_gamma, max_steps, N, cascade, 128, M, ptr(keep[3]),
             ptr(keep[4]), ptr(xt), ptr(dt_), ptr(lt), ptr(rt), ptr(ctr), ptr(keep[5]), ptr(ws), stream())
        tag = (seed, N, cascade, bound, max_steps, dt_gamma, M, total)
        assert np.array_equal(host(rt), ro_), tag
        assert np.array_equal(host(xt), xo) and np.array_equal(host(dt_), do) and np.array_equal(host(lt), lo), tag
        assert host(ctr).tolist() == [total, N], tag
        gx, gd = rng.normal(size=(M, 3)).astype(np.float32), rng.normal(size=(M, 3)).astype(np.float32)
        go, gdd = O.march_rays_train_backward(gx, gd, ro_, lo)
        g_o, g_d = torch.zeros(N, 3, device="cuda"), torch.zeros(N, 3, device="cuda")
        gxt, gdt = dev(gx), dev(gd)                  # raw pointers: the tensors must outlive the launch
        call("lz_march_rays_train_backward", ptr(gxt), ptr(gdt), ptr(rt), ptr(lt), N, M, ptr(g_o), ptr(g_d), stream())
        assert np.array_equal(host(g_o), go) and np.array_equal(host(g_d), gdd), tag


@pytest.mark.parametrize("seed", range(NCASE))
def test_inference_step_on_random_chunks(seed):
    """march_rays -> composite_rays_triplane (raymarching.py:347-398, 594-671) for random (n_alive, n_step) chunks of a partly finished
    frame: sample rows, padding, the in-place rays_alive / rays_t update and every accumulator bit for bit"""
    from lzzx_nerf_amd import raymarching as R
    rng = np.random.default_rng(10000 + seed)
    cascade = int(rng.choice([1, 1, 2]))
    bound = 1.0 if cascade == 1 else float(rng.choice([1.5, 2.0]))
    N = int(rng.choice([7, 64, 300, 1025, 4000]))
    ro, rd = _rays(rng, N, bound)
    aabb = (np.array([-1, -1, -1, 1, 1, 1], np.float32) * np.float32(bound)).astype(np.float32)
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.05)
    bits = _bitfield(rng, cascade)
    n_alive = int(rng.integers(1, N + 1))
    n_step = int(rng.choice([1, 2, 3, 5, 8]))
    max_steps = int(rng.choice([16, 64, 1024]))
    dt_gamma = float(rng.choice([0.0, 1 / 256]))
    T_thresh = float(rng.choice([1e-4, 1e-2, 0.5]))
    alive = rng.permutation(N)[:n_alive].astype(np.int32)
    span = np.where(fars < 1e30, fars - nears, 0).astype(np.float32)
    rays_t = (nears + rng.uniform(0, 1.1, N).astype(np.float32) * span).astype(np.float32)       # some already behind far
    xo, do, lo = O.march_rays(n_alive, n_step, alive, rays_t, ro, rd, bound, bits, cascade, 128, nears, fars, 128, None, dt_gamma, max_steps)
    xg, dg, lg = R.march_rays(n_alive, n_step, dev(alive), dev(rays_t), dev(ro), dev(rd), bound, dev(bits), cascade, 128, dev(nears), dev(fars),
                              128, False, dt_gamma, max_steps)
    tag = (seed, N, n_alive, n_step, cascade, max_steps)
    assert xg.shape == xo.shape and np.array_equal(host(xg), xo) and np.array_equal(host(dg), do) and np.array_equal(host(lg), lo), tag
    M = xo.shape[0]
    sig = (rng.uniform(0, 1, M) ** 3 * rng.choice([5.0, 60.0, 400.0])).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    a0, a1, unc = [rng.uniform(0, 1, M).astype(np.float32) for _ in range(3)]
    acc = {k: rng.uniform(0, 0.3, N).astype(np.float32) for k in ("ws", "dep", "a0s", "a1s", "us")}
    acc["img"] = rng.uniform(0, 0.3, (N, 3)).astype(np.float32)
    o = {k: v.copy() for k, v in acc.items()}
    al_o, rt_o = alive.copy(), rays_t.copy()
    O.composite_rays("triplane", n_alive, n_step, al_o, rt_o, sig, rgb, lo, o["ws"], o["dep"], o["img"], a0, a1, unc, o["a0s"], o["a1s"], o["us"],
                     T_thresh=T_thresh)
    g = {k: dev(v) for k, v in acc.items()}
    al_g, rt_g = dev(alive), dev(rays_t)
    R.composite_rays_triplane(n_alive, n_step, al_g, rt_g, dev(sig), dev(rgb), lg, dev(a0), dev(a1), dev(unc), g["ws"], g["dep"], g["img"],
                              g["a0s"], g["a1s"], g["us"], T_thresh)
    assert np.array_equal(host(al_g), al_o) and np.array_equal(host(rt_g), rt_o), tag
    for k in o:
        assert np.array_equal(host(g[k]), o[k]), (tag, k)


@pytest.mark.parametrize("seed", range(NCASE))
def test_composite_train_on_random_rays(seed):
    """composite_rays_train_triplane forward + backward (raymarching.py:594-671) on random ray tables: empty rays, rays longer than one
    staging chunk, a ray that overflows the sample buffer, T_thresh cuts"""
    from lzzx_nerf_amd import raymarching as R
    rng = np.random.default_rng(11000 + seed)
    N = int(rng.choice([1, 63, 64, 65, 300, 1500]))
    max_c = int(rng.choice([3, 40, 200]))
    counts = rng.integers(0, max_c, N)
    counts[rng.integers(0, N, max(N // 10, 1))] = 0
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    rays = np.stack([rng.permutation(N), offs, counts], 1).astype(np.int32)
    M = int(counts.sum()) + int(rng.integers(0, 130))
    if N > 2 and rng.random() < 0.5:
        rays[-1, 2] = M + 5                                   # overflows M: treated as empty (raymarching.cu:1904)
    M = max(M, 1)
    sig = (rng.uniform(0, 1, M) ** 2 * rng.choice([10.0, 80.0, 600.0])).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    dl = np.stack([rng.uniform(0.005, 0.03, M), np.cumsum(rng.uniform(0.01, 0.03, M)) + 2], 1).astype(np.float32)
    a0, a1, unc = [rng.uniform(0, 1, M).astype(np.float32) for _ in range(3)]
    T_thresh = float(rng.choice([1e-4, 1e-2, 0.3]))
    t = lambda a: dev(a).requires_grad_(True)
    sg, rg, a0g, a1g, ug = t(sig), t(rgb), t(a0), t(a1), t(unc)
    ws, a0s, a1s, us, dep, img = R.composite_rays_train_triplane(sg, rg, a0g, a1g, ug, dev(dl), dev(rays), T_thresh)
    fo = O.composite_rays_train_forward("triplane", sig, rgb, dl, rays, a0, a1, unc, T_thresh)
    valid = (rays[:, 2] > 0) & (rays[:, 1] + rays[:, 2] <= M)   # rays the kernel writes (the others keep the caller's buffer contents)
    ix = rays[valid, 0]
    for got, want in ((ws, "weights_sum"), (a0s, "amb0_sum"), (a1s, "amb1_sum"), (us, "unc_sum"), (dep, "depth"), (img, "image")):
        assert np.array_equal(host(got)[ix], fo[want][ix]), (seed, want)
    gws, ga0, ga1, gu, gd = [rng.normal(size=N).astype(np.float32) for _ in range(5)]
    gimg = rng.normal(size=(N, 3)).astype(np.float32)
    fwd = {k: host(v) for k, v in (("weights_sum", ws), ("amb0_sum", a0s), ("unc_sum", us), ("image", img))}
    go = O.composite_rays_train_backward("triplane", dict(grad_weights_sum=gws, grad_amb0_sum=ga0, grad_amb1_sum=ga1, grad_unc_sum=gu, grad_image=gimg),
                                         sig, rgb, dl, rays, fwd, a0, a1, unc, T_thresh)
    torch.autograd.backward([ws, a0s, a1s, us, dep, img], [dev(x) for x in (gws, ga0, ga1, gu, gd, gimg)])
    assert np.array_equal(host(sg.grad), go["grad_sigmas"]) and np.array_equal(host(rg.grad), go["grad_rgbs"]), seed
    assert np.array_equal(host(a0g.grad), go["grad_amb0"]) and np.array_equal(host(a1g.grad), go["grad_amb1"]), seed
    assert np.array_equal(host(ug.grad), go["grad_unc"]), seed


@pytest.mark.parametrize("seed", range(max(NCASE // 2, 1)))
def test_fused_head_on_random_batches(params, golden, seed):
    """the MFMA head (network.py:252-311 as one kernel) at batch sizes around its 16-sample slices and 512-sample workgroup tiles,
    positions on and beyond the bound, scaled audio / eye conditions: every output bit for bit against the checker"""
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from oracle.head import TriplaneSpec, head_forward
    rng = np.random.default_rng(12000 + seed)
    M = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 511, 512, 513, 1000, 4099]))
    testing = bool(rng.random() < 0.5)
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0)
    xyz = (rng.uniform(-1, 1, (M, 3)) * rng.choice([1.0, 1.0, 1.3])).astype(np.float32)
    xyz[rng.integers(0, M)] = rng.choice([-1.0, 0.0, 1.0], 3)
    d = rng.normal(size=(M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    enc_a = (golden["net_enc_a"] * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
    eye = (golden["net_eye"] * np.float32(rng.uniform(0.0, 1.5))).astype(np.float32)      # (these weights are an exp_eye network: sigma_net takes the eye feature)
    ind = golden["net_ind"]
    want = head_forward(TriplaneSpec(1.0), params, xyz, d, enc_a, ind, eye, testing=testing)
    got = head.forward(dev(xyz), dev(d), dev(enc_a), dev(ind), dev(eye), testing=testing)
    for name, g, w in zip(("sigma", "rgb", "amb_aud", "amb_eye", "unc"), got, want):
        if w is None:
            continue
        assert np.array_equal(host(g).reshape(-1), np.asarray(w).reshape(-1)), (seed, M, testing, name)
