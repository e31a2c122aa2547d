"""The STEP-MAJOR sample layout of the training path (include/lzzx_nerf_hip.h: lz_march_rays_train_grouped; raymarching.py: layout="step").

The reference leaves the row order of march_rays_train to its atomics (raymarching.cu:446-454): rays[] = (ray id, offset, count) is the
only statement of which rows a ray owns, and the reference layout keeps a ray's rows consecutive.  The step-major layout is a different
admissible arrangement of the SAME samples, so everything here is held to the ray-major operators (themselves bit-pinned to the CPU
checker, test_gpu_parity.py) through the row map of the header:

    row(j, k) = o_g + sum_i min(c_i, k) + #{ i < j : c_i > k }        (group g = rays[] rows G g .. G g + G - 1, G = lz_train_group_size())

* march: same (ray id, count) set, offsets = exclusive scan in processing order, every sample's xyz / dir / (dt, t) bit-equal at its mapped
  row, unowned rows zero -- with and without a sort order, with a non-zero counter base, with rays dropped for lack of room; and against
  the CPU checker's march directly;
* compositing forward / backward, all training variants: per-ray outputs and per-sample gradients bit-equal to the ray-major kernels on
  the permuted inputs (early termination included);
* the march's backward (train_camera): bit-equal;
* a whole training step (march -> fused head -> compositing -> loss -> backward): image and loss bit-equal, weight / table gradients to
  their float-atomic order."""
import numpy as np
import pytest
import torch

from conftest import make_params, synthetic_camera

pytestmark = pytest.mark.gpu

MAX_STEPS = 64


def group_size():
    from lzzx_nerf_amd import _lib
    return int(_lib.load().lz_train_group_size())


def step_rows(rays, M):
    """ray-major row -> step-major row for every sample a ray owns: dict-free numpy restatement of the header's formula.
    rays [N, 3] (id, ray-major offset, count) in processing order; returns (src, dst): sample k of rays[i] sits at ray-major row src and
    step-major row dst (dropped rays own nothing)."""
    rays = np.asarray(rays, np.int64)
    src, dst = [], []
    G = group_size()
    for g0 in range(0, len(rays), G):
        grp = rays[g0:g0 + G]
        c = np.where(grp[:, 1] + grp[:, 2] <= M, grp[:, 2], 0)
        if c.max(initial=0) == 0:
            continue
        gb = grp[0, 1]
        alive = c[None, :] > np.arange(c.max())[:, None]                 # [k, j]
        pos = np.cumsum(alive.reshape(-1)).reshape(alive.shape) - 1       # k-major running index
        k, j = np.nonzero(alive)
        dst.append(gb + pos[k, j])
        src.append(grp[j, 1] + k)
    if not src:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(src), np.concatenate(dst)


def _scene(name, device):
    from lzzx_nerf_amd.synthetic import ellipsoid_bitfield_device, ones_bitfield
    if name == "ones":
        return torch.from_numpy(ones_bitfield()).to(device)
    return ellipsoid_bitfield_device(device)[0]


def _rays(device, n_rays, size=96, seed=0):
    from lzzx_nerf_amd.utils import frame_rays
    pose, intr = synthetic_camera(size, size)
    ro, rd = frame_rays(torch.from_numpy(np.ascontiguousarray(pose)).to(device), intr, size, size)
    g = torch.Generator(device=device).manual_seed(seed)
    sel = torch.randperm(size * size, device=device, generator=g)[:n_rays]
    return ro[sel].contiguous(), rd[sel].contiguous()


def _march(ro, rd, bits, layout, order=None, mean_count=-1, force_all=True, ctr0=(0, 0), perturb=False):
    from lzzx_nerf_amd import raymarching as R
    aabb = torch.tensor([-1, -0.5, -1, 1, 0.5, 1], dtype=torch.float32, device=ro.device)
    nears, fars = R.near_far_from_aabb(ro.detach(), rd.detach(), aabb, 0.05)     # (no backward, like the reference's: raymarching.py:18-60)
    ctr = torch.tensor(ctr0, dtype=torch.int32, device=ro.device)
    out = R.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, ctr, mean_count, perturb, 128, force_all, 1 / 256, MAX_STEPS,
                             layout=layout, order=order)
    return out + (ctr,)


def _check_march(ray_major, step_major, M_rows, order):
    xr, dr, tr, rr, cr = [t.cpu().numpy() for t in ray_major]
    xs, ds, ts, rs, cs = [t.cpu().numpy() for t in step_major]
    assert (cr == cs).all() and xr.shape == xs.shape
    N = len(rr)
    assert (rr[:, 0] == np.arange(N)).all()
    if order is not None:
        assert (rs[:, 0] == order).all()
    assert sorted(rs[:, 0].tolist()) == list(range(N))
    assert (rs[:, 2] == rr[rs[:, 0], 2]).all()                                             # the same count per ray id
    assert (rs[:, 1] == rs[0, 1] + np.concatenate([[0], np.cumsum(rs[:-1, 2])])).all()   # exclusive scan in processing order
    src, dst = step_rows(rs, M_rows)
    srt = np.argsort(src, kind="stable")          # ascending processing-order rows = rays in processing order, steps in order inside a ray
    src, dst = src[srt], dst[srt]
    kept_c = np.where(rs[:, 1] + rs[:, 2] <= M_rows, rs[:, 2], 0)
    ids, offs = np.repeat(rs[:, 0], kept_c), np.repeat(rs[:, 1], kept_c)
    rm = rr[ids, 1] + (src - offs)                # the ray-major row of the same sample (sample k of ray id n: rr[n].offset + k)
    sel = rr[ids, 1] + rr[ids, 2] <= M_rows       # (the ray-major march may have dropped other rays when the orders differ)
    assert len(np.unique(dst)) == len(dst) and (dst < M_rows).all()
    assert np.array_equal(xs[dst[sel]], xr[rm[sel]]) and np.array_equal(ds[dst[sel]], dr[rm[sel]]) and np.array_equal(ts[dst[sel]], tr[rm[sel]])
    owned = np.zeros(len(xs), bool)
    owned[dst] = True
    assert not xs[~owned].any() and not ds[~owned].any() and not ts[~owned].any()
    return int(sel.sum()), int(len(sel))


@pytest.mark.parametrize("scene", ["ones", "ellipsoid"])
@pytest.mark.parametrize("order_kind", ["auto", "none", "random"])
def test_grouped_march_holds_the_ray_major_samples(scene, order_kind):
    dev = torch.device("cuda")
    bits = _scene(scene, dev)
    ro, rd = _rays(dev, 3000 + 37)              # not a multiple of 64: a ragged last group
    N = ro.shape[0]
    from lzzx_nerf_amd import raymarching as R
    if order_kind == "auto":
        order, expect = None, R.ray_order(ro, rd, 1.0).cpu().numpy()
    elif order_kind == "none":
        order, expect = False, np.arange(N)
    else:
        expect = np.random.default_rng(1).permutation(N).astype(np.int32)
        order = torch.from_numpy(expect).to(dev)
    a = _march(ro, rd, bits, "ray")
    b = _march(ro, rd, bits, "step", order=order)
    assert getattr(b[3], "lz_layout") == "step" and getattr(a[3], "lz_layout") == "ray"
    kept, total = _check_march(a, b, a[0].shape[0], expect)
    assert kept == total == int(a[4][0].item()) and kept > N        # nothing dropped, several samples per ray
    if order_kind == "auto":     # the locality order: neighbours in the order are neighbours on the image plane (mean step of the
        d = rd[torch.from_numpy(expect).long().to(dev)]            # direction between consecutive rays far below a random order's)
        near = float((d[1:] - d[:-1]).norm(dim=1).mean())
        rnd = float((rd[1:] - rd[:-1]).norm(dim=1).mean())
        assert near < 0.25 * rnd


def test_grouped_march_with_counter_base_and_dropped_rays():
    """the counter accumulates like the reference's atomicAdd (rows in front of the base stay zero) and a buffer too small for the step
    drops a SUFFIX of the processing order (raymarching.cu:457), whose rows read zero"""
    dev = torch.device("cuda")
    bits = _scene("ones", dev)
    ro, rd = _rays(dev, 2048 + 5)
    full = _march(ro, rd, bits, "ray")
    total = int(full[4][0].item())
    small = (total * 2 // 3) // 128 * 128            # steady-state buffer sized below the step's need
    a = _march(ro, rd, bits, "ray", mean_count=small - 128, force_all=False, ctr0=(256, 0))
    b = _march(ro, rd, bits, "step", order=False, mean_count=small - 128, force_all=False, ctr0=(256, 0))
    assert a[0].shape[0] == b[0].shape[0] == small
    rs = b[3].cpu().numpy()
    assert rs[0, 1] == 256 and int(b[4][0].item()) == 256 + total and int(b[4][1].item()) == len(rs)
    kept, tot = _check_march(a, b, small, np.arange(len(rs)))
    assert 0 < kept == tot < total                  # order=False: same processing order as ray-major, so the same suffix is dropped
    assert not b[0][:256].any()


def test_grouped_march_equals_the_checker():
    """directly against the CPU restatement of raymarching.cu:342-517 (ray-major there), through the row map"""
    from oracle import oracle as O
    dev = torch.device("cuda")
    bits = _scene("ellipsoid", dev)
    ro, rd = _rays(dev, 1000)
    b = _march(ro, rd, bits, "step")
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    nf = O.near_far_from_aabb(ro.cpu().numpy(), rd.cpu().numpy(), aabb, 0.05)
    M = b[0].shape[0]
    oxyz, odir, odl, orays = O.march_rays_train(ro.cpu().numpy(), rd.cpu().numpy(), 1.0, bits.cpu().numpy(), 1, 128, nf[0], nf[1], None, -1, None, 128,
                                                True, 1 / 256, MAX_STEPS)
    assert oxyz.shape[0] == M
    rs = b[3].cpu().numpy()
    src, dst = step_rows(rs, M)
    srt = np.argsort(src, kind="stable")
    ids = np.repeat(rs[:, 0], rs[:, 2])
    kk = src[srt] - np.repeat(rs[:, 1], rs[:, 2])
    orow = {int(r[0]): r for r in orays}              # the checker's rows are in ITS ray order: look a ray up by id
    oo = np.array([orow[int(n)][1] for n in rs[:, 0]]), np.array([orow[int(n)][2] for n in rs[:, 0]])
    assert (oo[1] == rs[:, 2]).all()
    rm = np.repeat(oo[0], rs[:, 2]) + kk
    assert len(dst) == int(rs[:, 2].sum()) > len(rs)
    assert np.array_equal(b[0].cpu().numpy()[dst[srt]], oxyz[rm]) and np.array_equal(b[1].cpu().numpy()[dst[srt]], odir[rm])
    assert np.array_equal(b[2].cpu().numpy()[dst[srt]], odl[rm])


def _permute_to_step(rs, M, *arrays):
    """ray-major sample arrays (in PROCESSING order offsets, i.e. rays rs themselves ray-major) -> step-major rows"""
    src, dst = step_rows(rs.cpu().numpy(), M)
    src_t, dst_t = torch.from_numpy(src).to(rs.device), torch.from_numpy(dst).to(rs.device)
    out = []
    for a in arrays:
        b = torch.zeros_like(a)
        b[dst_t] = a[src_t]
        out.append(b)
    return out, src_t, dst_t


VARIANTS = {"ambient": (1, 0, 0), "sigma": (1, 1, 0), "uncertainty": (1, 0, 1), "triplane": (2, 0, 1)}


@pytest.mark.parametrize("fits", [False, True])
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_grouped_compositing_equals_ray_major(variant, fits, monkeypatch):
    """forward and backward of every training variant on the step-major rows == the ray-major kernels on the same samples, bit for bit;
    sigma large enough that most rays stop early (T < T_thresh), a few rays dropped (buffer shorter than the step)"""
    from lzzx_nerf_amd import raymarching as R
    dev = torch.device("cuda")
    bits = _scene("ones", dev)
    ro, rd = _rays(dev, 1500 + 11)
    xyzs, dirs, deltas, rays, ctr = _march(ro, rd, bits, "ray")
    total = int(ctr[0].item())
    M = (total + 300) if fits else (total - 900) // 128 * 128       # padding rows behind the last ray / the last rays do not fit: dropped by both layouts
    g = torch.Generator(device=dev).manual_seed(3)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
    sig, rgb, a0, a1, un = rnd(M) * 40, rnd(M, 3), rnd(M), rnd(M), rnd(M)
    dl = torch.zeros(M, 2, device=dev)
    dl[:min(M, deltas.shape[0])] = deltas[:M]
    na, aw, hu = VARIANTS[variant]
    N = rays.shape[0]
    (sig_s, rgb_s, a0_s, a1_s, un_s, dl_s), src, dst = _permute_to_step(rays, M, sig, rgb, a0, a1, un, dl)
    f_r = R._composite_train_fwd((na, aw, hu), sig, rgb, a0, a1 if na > 1 else None, un if hu else None, dl, rays, 1e-4, 0)
    f_s = R._composite_train_fwd((na, aw, hu), sig_s, rgb_s, a0_s, a1_s if na > 1 else None, un_s if hu else None, dl_s, rays, 1e-4, 1)
    for x, y in zip(f_r, f_s):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y)
    assert float(f_r[0].max()) > 0.99                     # early termination happened
    gws, ga0, ga1, gu, gim = rnd(N), rnd(N), rnd(N), rnd(N), rnd(N, 3)
    ws, a0s, a1s, us, dep, img = f_r
    # the step-major backward writes EVERY row of the gradient buffers itself (the wrapper hands it torch.empty_like): poison what it gets
    monkeypatch.setattr(torch, "empty_like", lambda t, **kw: torch.full_like(t, float("nan")))
    b_r = R._composite_train_bwd((na, aw, hu), gws, ga0, ga1 if na > 1 else None, gu if hu else None, gim, sig, rgb, a0, a1 if na > 1 else None,
                                 un if hu else None, dl, rays, ws, a0s, us, img, 1e-4, 0)
    b_s = R._composite_train_bwd((na, aw, hu), gws, ga0, ga1 if na > 1 else None, gu if hu else None, gim, sig_s, rgb_s, a0_s,
                                 a1_s if na > 1 else None, un_s if hu else None, dl_s, rays, ws, a0s, us, img, 1e-4, 1)
    for x, y in zip(b_r, b_s):
        assert (x is None) == (y is None)
        if x is not None:
            z = torch.zeros_like(x)
            z[dst] = x[src]
            assert torch.equal(z, y)
    assert float(b_r[0].abs().max()) > 0


def test_grouped_march_backward_equals_ray_major():
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd._util import call, ptr, stream
    dev = torch.device("cuda")
    bits = _scene("ellipsoid", dev)
    ro, rd = _rays(dev, 777)
    xyzs, dirs, deltas, rays, ctr = _march(ro, rd, bits, "ray")
    M, N = xyzs.shape[0], rays.shape[0]
    g = torch.Generator(device=dev).manual_seed(5)
    gx, gd = torch.randn(M, 3, device=dev, generator=g), torch.randn(M, 3, device=dev, generator=g)
    (gx_s, gd_s, dl_s), _, _ = _permute_to_step(rays, M, gx, gd, deltas)
    outs = []
    for name, a, b, c in (("lz_march_rays_train_backward", gx, gd, deltas), ("lz_march_rays_train_backward_grouped", gx_s, gd_s, dl_s)):
        go, gdd = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
        call(name, ptr(a), ptr(b), ptr(rays), ptr(c.contiguous()), N, M, ptr(go), ptr(gdd), stream())
        outs.append((go, gdd))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and float(outs[0][0].abs().max()) > 0


def test_training_step_is_layout_independent():
    """march -> fused -O head -> compositing -> loss -> backward under both layouts: the image and the loss are the same bits (per-sample
    head outputs do not depend on a sample's neighbours, per-ray compositing is the same arithmetic), gradients agree to the order of their
    float atomics / per-workgroup partial sums"""
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd.head_train import FusedTriplaneTrainHead
    dev = torch.device("cuda")
    golden = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "reference_python.npz"))
    P = make_params(golden)
    cond = [torch.from_numpy(golden[k]).to(dev) for k in ("net_enc_a", "net_ind", "net_eye")]
    bits = _scene("ones", dev)
    ro, rd = _rays(dev, 4096)
    target = torch.rand(4096, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    res = {}
    for layout in ("ray", "step"):
        net = FusedTriplaneTrainHead(P, bound=1.0, forward_dtype="f16", backward_dtype="f16").to(dev)
        xyzs, dirs, deltas, rays, _ = _march(ro, rd, bits, layout)
        sigma, rgb, a0, a1, unc = net(xyzs, dirs, *cond)
        ws, a0s, a1s, us, dep, img = R.composite_rays_train_triplane(sigma, rgb, a0.squeeze(-1), a1.squeeze(-1), unc.squeeze(-1), deltas, rays)
        loss = ((img + (1 - ws).unsqueeze(-1) - target) ** 2).mean() + 1e-4 * a0s.mean() + 1e-4 * a1s.mean() + 1e-3 * us.mean()
        (loss * 1024.0).backward()
        res[layout] = (loss.detach(), img.detach(), ws.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    assert torch.equal(res["ray"][0], res["step"][0]) and torch.equal(res["ray"][1], res["step"][1]) and torch.equal(res["ray"][2], res["step"][2])
    for k, ga in res["ray"][3].items():
        gb = res["step"][3][k]
        scale = float(ga.abs().max())
        assert scale > 0 and torch.isfinite(gb).all(), k
        assert float((ga - gb).abs().max()) <= 2e-3 * scale, (k, float((ga - gb).abs().max()), scale)


def test_a_callers_order_must_be_a_permutation():
    dev = torch.device("cuda")
    bits = _scene("ones", dev)
    ro, rd = _rays(dev, 130)
    bad = torch.arange(130, dtype=torch.int32, device=dev)
    bad[7] = 8                                             # a duplicate (and 7 missing)
    with pytest.raises(ValueError, match="permutation"):
        _march(ro, rd, bits, "step", order=bad)
    bad = torch.arange(130, dtype=torch.int32, device=dev)
    bad[0] = 130                                           # out of range
    with pytest.raises(ValueError, match="permutation"):
        _march(ro, rd, bits, "step", order=bad)
    with pytest.raises(ValueError, match="entries"):
        _march(ro, rd, bits, "step", order=torch.arange(64, dtype=torch.int32, device=dev))
    with pytest.raises(ValueError, match="layout"):
        _march(ro, rd, bits, "steps")


@pytest.mark.parametrize("n_rays", [1, 63, 64, 65, 129, 1000])
@pytest.mark.parametrize("perturb", [False, True])
def test_ragged_groups_empty_rays_and_perturbed_starts(n_rays, perturb):
    """group edges and rays without a sample: fewer rays than a group, one ray more than a group, rays that miss the occupied cells
    (count 0 inside a group), perturbed start times (the same noise under both layouts: the generator is re-seeded) -- march rows and
    the triplane compositing forward / backward against the ray-major operators"""
    from lzzx_nerf_amd import raymarching as R
    dev = torch.device("cuda")
    bits = _scene("ellipsoid", dev)
    ro, rd = _rays(dev, n_rays, size=64, seed=n_rays)
    torch.manual_seed(1234)
    a = _march(ro, rd, bits, "ray", perturb=perturb)
    torch.manual_seed(1234)
    b = _march(ro, rd, bits, "step", perturb=perturb)
    counts = a[3][:, 2].cpu().numpy()
    if n_rays >= 63:
        assert (counts == 0).any() and (counts > 0).any()          # the 64 x 64 frame of the ellipsoid scene has rays of both kinds
    _check_march(a, b, a[0].shape[0], None)
    M, N = a[0].shape[0], n_rays
    if M == 0:
        return
    g = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
    sig, rgb, a0, a1, un = rnd(M) * 30, rnd(M, 3), rnd(M), rnd(M), rnd(M)
    # the SAME per-sample values under both layouts: ray-major values -> step-major rows through the two rays tables
    ra, rb = a[3].cpu().numpy(), b[3].cpu().numpy()
    src_b, dst_b = step_rows(rb, M)                                   # processing-order ray-major row -> step row
    srt = np.argsort(src_b, kind="stable")
    ids = np.repeat(rb[:, 0], rb[:, 2])
    kk = src_b[srt] - np.repeat(rb[:, 1], rb[:, 2])
    rm = torch.from_numpy(ra[ids, 1] + kk).to(dev)                    # the sample's row in the ray-major buffers
    st = torch.from_numpy(dst_b[srt]).to(dev)
    perm = lambda x: torch.zeros_like(x).index_copy_(0, st, x[rm])
    f_r = R._composite_train_fwd((2, 0, 1), sig, rgb, a0, a1, un, a[2].contiguous(), a[3], 1e-4, 0)
    f_s = R._composite_train_fwd((2, 0, 1), perm(sig), perm(rgb), perm(a0), perm(a1), perm(un), b[2].contiguous(), b[3], 1e-4, 1)
    for x, y in zip(f_r, f_s):
        assert torch.equal(x, y)
    gws, ga0, ga1, gu, gim = rnd(N), rnd(N), rnd(N), rnd(N), rnd(N, 3)
    ws, a0s, a1s, us, dep, img = f_r
    b_r = R._composite_train_bwd((2, 0, 1), gws, ga0, ga1, gu, gim, sig, rgb, a0, a1, un, a[2].contiguous(), a[3], ws, a0s, us, img, 1e-4, 0)
    b_s = R._composite_train_bwd((2, 0, 1), gws, ga0, ga1, gu, gim, perm(sig), perm(rgb), perm(a0), perm(a1), perm(un), b[2].contiguous(), b[3], ws,
                                 a0s, us, img, 1e-4, 1)
    for x, y in zip(b_r, b_s):
        assert torch.equal(perm(x), y)


def test_camera_gradients_through_the_step_major_march():
    """opt.train_camera (renderer.py:225-230): rays that require gradients get them through march_rays_train's backward under either layout
    -- the same sums over a ray's samples, in step order along the ray, so the same bits"""
    dev = torch.device("cuda")
    bits = _scene("ellipsoid", dev)
    ro, rd = _rays(dev, 300)
    out = {}
    for layout in ("ray", "step"):
        o, d = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
        xyzs, dirs, deltas, rays, _ = _march(o, d, bits, layout)
        w = torch.linspace(0.5, 1.5, 3, device=dev)
        # a per-ray-separable loss (every sample of ray n weighted by a function of n): independent of which rows the samples sit in
        ids = torch.zeros(xyzs.shape[0], dtype=torch.long, device=dev)
        from lzzx_nerf_amd import raymarching as R
        if layout == "ray":
            r = rays.long()
            ids[:int(r[:, 2].sum())] = torch.repeat_interleave(r[:, 0], r[:, 2])
        else:
            src, dst = step_rows(rays.cpu().numpy(), xyzs.shape[0])
            rs = rays.cpu().numpy()
            srt = np.argsort(src, kind="stable")
            ids[torch.from_numpy(dst[srt]).to(dev)] = torch.from_numpy(np.repeat(rs[:, 0], rs[:, 2])).to(dev).long()
        scale = (1.0 + 0.01 * ids.float()).unsqueeze(1)
        ((xyzs * w * scale).sum() + (dirs * scale).sum() * 0.25).backward()
        out[layout] = (o.grad.clone(), d.grad.clone())
    assert torch.equal(out["ray"][0], out["step"][0]) and torch.equal(out["ray"][1], out["step"][1])
    assert float(out["ray"][0].abs().max()) > 0 and float(out["ray"][1].abs().max()) > 0
