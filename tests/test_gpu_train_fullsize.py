"""BASELINE cfg3 at FULL size under -m gpu (Missing 4): 65 536 random rays of the 512x512 frame, max_steps 192, both scenes,
through march_rays_train -> FusedTriplaneTrainHead -> composite_rays_train_triplane -> loss -> backward, i.e. the step bench.py times
(reference: renderer.py:279-304, raymarching.py:186-280, raymarching.cu:1999-2122, gridencoder.cu:226-313).  The gradient tests of
test_gpu_train_step.py run 24^2 / 48^2 rays against a float64 model; nothing there reaches the lane-major row walk of the table scatter (full 73 728-row chunks), the
register-keep path of the LDS grid backward or the mean_count-sized buffers.  A float64 model of 6 M samples is out of reach, so the step
is held to size-independent properties:

* the sample buffer does not overflow and counter[0] = sum of the per-ray counts = the samples the INFERENCE march (march_rays, one
  n_step = max_steps chunk) finds on the same rays;
* the loss agrees across the four arrangements (recorded f32, recomputing, half records, all-half) within their stated tolerances;
* every gradient is finite and non-zero;
* the table scatter is a partition of unity: per plane and level, the sum of grad_embeddings equals the sum of the incoming feature
  gradient (bilinear corner weights sum to 1);
* the forward's bits are identical run to run;
* and the scatter itself equals the CPU checker's (oracle.grid_encode_backward) on the 4 096-ray prefix of the step's samples, fed the
  very feature gradient the GPU scatter consumed."""
import numpy as np
import pytest
import torch

from conftest import synthetic_camera
from oracle import oracle as O

pytestmark = pytest.mark.gpu

N_RAYS, MAX_STEPS, SIZE = 65536, 192, 512
ARRANGEMENTS = {"f32_records": dict(), "recompute": dict(record=False), "f16_records": dict(record_dtype="f16"),
                "all_f16": dict(forward_dtype="f16", backward_dtype="f16")}


def _scene(name, device):
    from lzzx_nerf_amd.synthetic import ellipsoid_bitfield_device, ones_bitfield
    if name == "ones":
        return torch.from_numpy(ones_bitfield()).to(device)
    return ellipsoid_bitfield_device(device)[0]


def _rays(device, n_rays=N_RAYS):
    from lzzx_nerf_amd.utils import frame_rays
    pose, intr = synthetic_camera(SIZE, SIZE)
    ro, rd = frame_rays(torch.from_numpy(np.ascontiguousarray(pose)).to(device), intr, SIZE, SIZE)
    g = torch.Generator(device=device).manual_seed(0)
    sel = torch.randperm(SIZE * SIZE, device=device, generator=g)[:n_rays]
    target = torch.rand(n_rays, 3, device=device, generator=g)
    return ro[sel].contiguous(), rd[sel].contiguous(), target


def _march(ro, rd, bits, mean_count=-1, force_all=True, layout="ray"):
    from lzzx_nerf_amd import raymarching as R
    aabb = torch.tensor([-1, -0.5, -1, 1, 0.5, 1], dtype=torch.float32, device=ro.device)
    nears, fars = R.near_far_from_aabb(ro, rd, aabb, 0.05)
    ctr = torch.zeros(2, dtype=torch.int32, device=ro.device)
    xyzs, dirs, deltas, rays = R.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, ctr, mean_count, False, 128, force_all, 1 / 256,
                                                  MAX_STEPS, layout=layout)
    return xyzs, dirs, deltas, rays, ctr, nears, fars


def _step(net, xyzs, dirs, deltas, rays, target, cond, scale=1.0):
    from lzzx_nerf_amd import raymarching as R
    enc_a, ind, eye = cond
    sigma, rgb, a0, a1, unc = net(xyzs, dirs, enc_a, ind, eye)
    ws, a0s, a1s, us, dep, img = R.composite_rays_train_triplane(sigma, rgb, a0.squeeze(-1), a1.squeeze(-1), unc.squeeze(-1), deltas, rays)
    loss = ((img + (1 - ws).unsqueeze(-1) - target) ** 2).mean() + 1e-4 * a0s.mean() + 1e-4 * a1s.mean() + 1e-3 * us.mean()
    (loss * scale).backward()
    return loss.detach(), (sigma.detach(), rgb.detach(), a0.detach(), a1.detach(), unc.detach())


@pytest.mark.parametrize("scene", ["ones", "ellipsoid"])
def test_full_size_march_counts(scene):
    """counter[0] = sum of the per-ray counts = rows returned; offsets tile [0, counter[0]) in atomic order without overlap; and the same
    rays marched by the INFERENCE kernel in one max_steps chunk yield the same count per ray (both walk the same occupied cells:
    raymarching.cu:342-470 vs 855-929); the steady-state buffers (mean_count from a first step, raymarching.py:221-256) hold the step"""
    from lzzx_nerf_amd import raymarching as R
    dev = torch.device("cuda")
    bits = _scene(scene, dev)
    ro, rd, _ = _rays(dev)
    xyzs, dirs, deltas, rays, ctr, nears, fars = _march(ro, rd, bits)
    M = int(ctr[0].item())
    rows = xyzs.shape[0]                      # padded to a multiple of 128 like the reference's buffers (raymarching.py:246-256)
    assert int(ctr[1].item()) == N_RAYS and M > 0 and M <= rows <= M + 128 and rows % 128 == 0
    assert not bool(deltas[M:].any())
    cnt, off = rays[:, 2].long(), rays[:, 1].long()
    assert int(cnt.sum()) == M and int(cnt.max()) <= MAX_STEPS
    order = torch.argsort(off[cnt > 0])
    o_s, c_s = off[cnt > 0][order], cnt[cnt > 0][order]
    assert int(o_s[0]) == 0 and bool((o_s[1:] == (o_s + c_s)[:-1]).all()) and int(o_s[-1] + c_s[-1]) == M
    assert sorted(rays[:, 0].tolist()) == list(range(N_RAYS))
    # inference march on the same rays: every ray alive, t = near, one chunk of max_steps samples
    alive = torch.arange(N_RAYS, dtype=torch.int32, device=dev)
    x2, d2, dl2 = R.march_rays(N_RAYS, MAX_STEPS, alive, nears.clone(), ro, rd, 1.0, bits, 1, 128, nears, fars, -1, False, 1 / 256, MAX_STEPS)
    inf_cnt = (dl2.view(-1, MAX_STEPS, 2)[:N_RAYS, :, 0] != 0).sum(1)
    per_ray = torch.zeros(N_RAYS, dtype=torch.long, device=dev)
    per_ray[rays[:, 0].long()] = cnt
    assert torch.equal(per_ray, inf_cnt.long())
    # the positions agree too: ray i's k-th training sample is the inference march's k-th row of that ray
    i = int(rays[123, 0]); o, c = int(rays[123, 1]), int(rays[123, 2])
    assert torch.equal(xyzs[o:o + c], x2.view(-1, MAX_STEPS, 3)[i, :c])
    del x2, d2, dl2
    # steady state: buffers sized from the first step's count plus bench.py's margin; nothing is dropped, same samples
    mean_count = M + N_RAYS // 64
    xs, ds, dls, rs, ctr2, _, _ = _march(ro, rd, bits, mean_count=mean_count, force_all=False)
    assert int(ctr2[0].item()) == M and xs.shape[0] >= M and int(ctr2[0].item()) <= mean_count
    per_ray2 = torch.zeros(N_RAYS, dtype=torch.long, device=dev)
    per_ray2[rs[:, 0].long()] = rs[:, 2].long()
    assert torch.equal(per_ray2, per_ray)


@pytest.mark.parametrize("scene", ["ones", "ellipsoid"])
def test_full_size_training_step_properties(params, golden, scene):
    from lzzx_nerf_amd.head_train import FusedTriplaneTrainHead
    dev = torch.device("cuda")
    bits = _scene(scene, dev)
    ro, rd, target = _rays(dev)
    cond = tuple(torch.from_numpy(np.ascontiguousarray(golden[k])).to(dev) for k in ("net_enc_a", "net_ind", "net_eye"))
    xyzs, dirs, deltas, rays, ctr, _, _ = _march(ro, rd, bits)
    M = int(ctr[0].item())
    losses, grads = {}, {}
    for name, arr in ARRANGEMENTS.items():
        net = FusedTriplaneTrainHead(dict(params), bound=1.0, **arr).to(dev)
        net.keep_denc = True
        half = name in ("f16_records", "all_f16")
        scale = 65536.0 if half else 1.0      # the GradScaler bench.py and the reference's trainer put in front of half operands
        loss, outs = _step(net, xyzs, dirs, deltas, rays, target, cond, scale)
        losses[name] = float(loss)
        g = {n: p.grad / scale for n, p in net.named_parameters()}
        assert len(g) == 14
        for n, t in g.items():
            assert bool(torch.isfinite(t).all()), (name, n)
            assert float(t.abs().max()) > 0, (name, n)
        grads[name] = {n: t.double().cpu() for n, t in g.items() if "embeddings" not in n}
        # partition of unity, per plane and level: sum of the table gradient = sum of the feature gradient the scatter was fed
        denc = net.last_denc.double() / scale                       # [3, 12, M]
        off = net.offsets.cpu().numpy().astype(np.int64)
        for p, enc in enumerate((net.encoder_xy, net.encoder_yz, net.encoder_xz)):
            ge = enc.embeddings.grad.double()[:, 0] / scale
            for l in range(12):
                want, got = float(denc[p, l].sum()), float(ge[off[l]:off[l + 1]].sum())
                ref = float(denc[p, l].abs().sum())
                assert abs(got - want) <= (2e-3 if half else 1e-4) * ref + 1e-12, (name, p, l, got, want, ref)
        if name == "f32_records":
            # run to run: the forward's bits, and the loss
            net2 = FusedTriplaneTrainHead(dict(params), bound=1.0).to(dev)
            loss2, outs2 = _step(net2, xyzs, dirs, deltas, rays, target, cond)
            for a, b in zip(outs, outs2):
                assert torch.equal(a, b)
            assert float(loss2) == float(loss)
            # the scatter against the checker on the prefix of samples that belongs to the first 4 096 rays in buffer order: the SAME
            # feature gradient, the checker's atomic-free accumulation (gridencoder.cu:226-313 restated in oracle/grid_oracle.c)
            order = torch.argsort(rays[:, 1].long())
            last = order[4095]
            m = int(rays[last, 1] + rays[last, 2])
            assert 0 < m <= M
            from lzzx_nerf_amd._util import call, ptr, stream
            x01 = torch.empty(3, m, 2, device=dev)                      # (x + bound) / (2 bound) per plane, exactly as the step maps them
            call("lz_triplane_plane_coords", ptr(xyzs[:m].contiguous()), m, 1.0, ptr(x01), stream())
            pls = float(np.exp2(np.float32(net.S)))
            for p, enc in enumerate((net.encoder_xy, net.encoder_yz, net.encoder_xz)):
                gt = net.last_denc[p, :, :m].contiguous()                 # [12, m] level-major: what the GPU scatter reads
                want, _ = O.grid_encode_backward(gt.t().contiguous().cpu().numpy(), x01[p].cpu().numpy(), (int(off[-1]), 1), off.astype(np.int32), pls, 64)
                # the GPU scatter over that prefix alone (the whole-step scatter holds every sample), same entry and gradient layout
                ge = torch.zeros(int(off[-1]), 1, device=dev)
                call("lz_grid_encode_backward", ptr(gt), ptr(x01[p]), ptr(enc.embeddings.detach()), ptr(net.offsets), ptr(ge), m, 2, 1, 12, net.S, 64,
                     None, None, 0, 0, 0, 3, stream())
                got = ge.cpu().numpy()
                tol = 1e-4 * float(np.abs(want).max())
                assert np.abs(got - want).max() <= tol, (p, float(np.abs(got - want).max()), tol)
        del net
        torch.cuda.empty_cache()
    # same loss: the f32 arrangements to rounding, the half ones within half precision of the outputs
    assert abs(losses["recompute"] - losses["f32_records"]) <= 1e-6 * abs(losses["f32_records"]), losses
    assert abs(losses["f16_records"] - losses["f32_records"]) <= 1e-6 * abs(losses["f32_records"]), losses     # half RECORDS leave the forward alone
    assert abs(losses["all_f16"] - losses["f32_records"]) <= 2e-3 * abs(losses["f32_records"]), losses
    # same weight gradients: recompute == record bit for bit on the wide layers is test_gpu_train_step's; here every arrangement against
    # the f32 one in the relative l2 norm of each matrix
    for name, tol in (("recompute", 1e-4), ("f16_records", 2e-2), ("all_f16", 6e-2)):
        for n, want in grads["f32_records"].items():
            got = grads[name][n]
            l2 = float((got - want).norm() / want.norm())
            assert l2 <= tol, (name, n, l2)


@pytest.mark.parametrize("scene", ["ones", "ellipsoid"])
def test_full_size_step_major_rows_give_the_same_step(params, golden, scene):
    """the bench's `-O` step on STEP-MAJOR sample rows (march_rays_train(layout="step"), tests/test_gpu_step_layout.py for the operators) at
    full size: same counts per ray, the image and the loss are the ray-major step's bits, the table scatter is still a partition of unity
    per plane and level (the lane-major row walk of lz_k_grid_backward_lds_fx), weight gradients agree in the relative l2 norm"""
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd.head_train import FusedTriplaneTrainHead
    dev = torch.device("cuda")
    bits = _scene(scene, dev)
    ro, rd, target = _rays(dev)
    cond = tuple(torch.from_numpy(np.ascontiguousarray(golden[k])).to(dev) for k in ("net_enc_a", "net_ind", "net_eye"))
    res = {}
    for layout in ("ray", "step"):
        xyzs, dirs, deltas, rays, ctr, _, _ = _march(ro, rd, bits, layout=layout)
        assert rays.lz_layout == layout
        net = FusedTriplaneTrainHead(dict(params), bound=1.0, forward_dtype="f16", backward_dtype="f16").to(dev)
        net.keep_denc = True
        enc_a, ind, eye = cond
        sigma, rgb, a0, a1, unc = net(xyzs, dirs, enc_a, ind, eye)
        ws, a0s, a1s, us, dep, img = R.composite_rays_train_triplane(sigma, rgb, a0.squeeze(-1), a1.squeeze(-1), unc.squeeze(-1), deltas, rays)
        loss = ((img + (1 - ws).unsqueeze(-1) - target) ** 2).mean() + 1e-4 * a0s.mean() + 1e-4 * a1s.mean() + 1e-3 * us.mean()
        (loss * 65536.0).backward()
        per_ray = torch.zeros(N_RAYS, dtype=torch.long, device=dev)
        per_ray[rays[:, 0].long()] = rays[:, 2].long()
        denc = net.last_denc.double() / 65536.0
        off = net.offsets.cpu().numpy().astype(np.int64)
        for p, enc in enumerate((net.encoder_xy, net.encoder_yz, net.encoder_xz)):
            ge = enc.embeddings.grad.double()[:, 0] / 65536.0
            assert bool(torch.isfinite(ge).all()) and float(ge.abs().max()) > 0
            for l in range(12):
                want, got, ref = float(denc[p, l].sum()), float(ge[off[l]:off[l + 1]].sum()), float(denc[p, l].abs().sum())
                assert abs(got - want) <= 2e-3 * ref + 1e-12, (layout, p, l, got, want, ref)
        res[layout] = (int(ctr[0].item()), per_ray, loss.detach(), img.detach(), ws.detach(),
                       {n: (q.grad / 65536.0).double().cpu() for n, q in net.named_parameters()})
        del net, sigma, rgb, a0, a1, unc, xyzs, dirs, deltas
        torch.cuda.empty_cache()
    a, b = res["ray"], res["step"]
    assert a[0] == b[0] and torch.equal(a[1], b[1])
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    for n, want in a[5].items():
        l2 = float((b[5][n] - want).norm() / want.norm())
        assert l2 <= 1e-3, (n, l2)
