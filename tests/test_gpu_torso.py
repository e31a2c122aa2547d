"""Torso branch: lzzx_nerf_amd.torso.FusedTorso (one kernel per frame) against the CPU restatement of
run_torso / forward_torso (oracle/torso.py) on seeded weights: masks, alpha, colour and deformation bit for bit."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle.torso import run_torso

pytestmark = pytest.mark.gpu
F32 = np.float32


def _torso_state(ind_dim, seed=0):
    rng = np.random.default_rng(seed)
    lin = lambda n, k: rng.uniform(-1, 1, (n, k)).astype(F32) / F32(np.sqrt(k))
    pls = np.exp2(np.log2(2048 / 16) / 15)
    offsets = O.grid_offsets(2, 16, pls, 16, 16)
    K0 = 34 + 42 + ind_dim
    sd = {"anchor_points": np.array([[0.01, 0.01, 0.1, 1], [-0.1, -0.1, 0.1, 1], [0.1, -0.1, 0.1, 1]], F32),
          "torso_deform_net.net.0.weight": lin(32, K0), "torso_deform_net.net.1.weight": lin(32, 32),
          "torso_deform_net.net.2.weight": lin(2, 32) * F32(0.2),
          "torso_net.net.0.weight": lin(32, 32 + K0), "torso_net.net.1.weight": lin(32, 32), "torso_net.net.2.weight": lin(4, 32),
          "torso_encoder.embeddings": rng.uniform(-1, 1, (int(offsets[-1]), 2)).astype(F32),
          "torso_encoder.offsets": offsets.astype(np.int32)}
    return sd


@pytest.mark.parametrize("ind_dim,masked", [(8, True), (0, True), (8, False)])
def test_fused_torso_matches_checker(ind_dim, masked):
    from lzzx_nerf_amd.torso import FusedTorso
    sd = _torso_state(ind_dim)
    rng = np.random.default_rng(5)
    H = W = 96
    ys, xs = np.meshgrid(np.linspace(-1, 1, H, dtype=F32), np.linspace(-1, 1, W, dtype=F32), indexing="ij")
    bg = np.stack([xs.ravel(), ys.ravel()], 1).astype(F32)                    # get_bg_coords: pixel grid in [-1, 1]
    bg[:5] = [[-1, -1], [1, 1], [1, -1], [0, 0], [0.999999, -0.999999]]
    G = 128
    yy, xx = np.meshgrid(np.arange(G), np.arange(G), indexing="ij")
    dens = np.exp(-(((xx - 64) / 30.0) ** 2 + ((yy - 80) / 40.0) ** 2)).astype(F32).reshape(-1)   # a torso-like blob
    pose = np.eye(4, dtype=F32)
    pose[:3, 3] = [0.05, -0.02, 3.3]
    th = 0.1
    pose[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], F32)
    ind = rng.normal(size=(1, ind_dim)).astype(F32) * F32(0.1) if ind_dim else None
    torso = FusedTorso({k: torch.from_numpy(v) for k, v in sd.items()}, torso_shrink=0.8)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    enc_anchor = torso.encode_anchor(dev(pose[None]))                          # torch + the frequency operator, caller side
    a_g, c_g, d_g = torso(dev(bg), ind_code=None if ind is None else dev(ind), density_grid=dev(dens) if masked else None,
                          density_thresh=0.05, enc_anchor=enc_anchor)
    a_o, c_o, d_o, mask = run_torso(sd, bg, enc_anchor.cpu().numpy().reshape(-1), ind, dens if masked else None, 0.05, 0.8)
    assert np.array_equal(a_g.cpu().numpy(), a_o) and np.array_equal(c_g.cpu().numpy(), c_o) and np.array_equal(d_g.cpu().numpy(), d_o)
    if masked:
        assert 0.05 < mask.mean() < 0.95 and (a_o[~mask] == 0).all()
    assert np.abs(d_o).max() > 1e-3 and a_o.max() > 0.3
    # mixing with the background (renderer.py:621) feeds the head renderer's bg_color
    mixed = FusedTorso.mix_background(a_g, c_g, 1.0)
    assert mixed.shape == (H * W, 3) and bool(((mixed >= -0.01) & (mixed <= 1.01)).all())


def test_torso_anchor_encoding_matches_numpy():
    """network.py:179-183 on the caller side: anchors @ inverse(pose^T), perspective divide, frequency encoding (deg 3)"""
    from lzzx_nerf_amd.torso import FusedTorso
    sd = _torso_state(8)
    torso = FusedTorso({k: torch.from_numpy(v) for k, v in sd.items()})
    pose = np.eye(4, dtype=F32)
    pose[:3, 3] = [0.1, 0.2, 3.0]
    e = torso.encode_anchor(torch.from_numpy(pose[None]).cuda()).cpu().numpy().reshape(-1)
    wa = sd["anchor_points"].astype(np.float64) @ np.linalg.inv(pose.T.astype(np.float64))
    wa = (wa[:, :2] / wa[:, 3:4] / wa[:, 2:3]).reshape(1, -1).astype(F32)
    ref = O.freq_encode_forward(wa, 3).reshape(-1)
    assert e.shape == (42,) and np.allclose(e, ref, atol=2e-5)


def test_full_frame_pipeline_matches_composed_checker(params, golden):
    """audio -> enc_a, torso -> background, head render over it (run_cuda order, renderer.py:406-570): every stage against its
    checker, composed the same way; frame bit for bit"""
    from conftest import ellipsoid_bitfield, synthetic_camera
    from lzzx_nerf_amd.pipeline import TalkingHeadFrame
    from oracle.audio import encode_audio
    from oracle.head import TriplaneSpec, get_rays
    from oracle.render import render_inference
    from test_audio_oracle import audio_state
    sd = dict(params)
    sd.update(_torso_state(8))
    sd.update(audio_state(29, 32, True))
    H = W = 48
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = ellipsoid_bitfield()[0]
    rng = np.random.default_rng(9)
    auds = rng.normal(size=(8, 29, 16)).astype(F32)
    ys, xs = np.meshgrid(np.linspace(-1, 1, H, dtype=F32), np.linspace(-1, 1, W, dtype=F32), indexing="ij")
    bg = np.stack([xs.ravel(), ys.ravel()], 1).astype(F32)
    ind_t = (rng.normal(size=(1, 8)) * 0.1).astype(F32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    frame = TalkingHeadFrame({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, dev(bits), bound=1.0)
    out = frame.render(dev(ro), dev(rd), dev(auds), eye=dev(golden["net_eye"]), ind_code=dev(golden["net_ind"]), bg_coords=dev(bg),
                       poses=dev(pose[None]), ind_code_torso=dev(ind_t), bg_color=1.0, max_steps=48)
    # checker, stage by stage
    enc_a = encode_audio(sd, auds, True)
    assert np.array_equal(out["enc_a"].cpu().numpy(), enc_a)
    enc_anchor = frame.torso.encode_anchor(dev(pose[None])).cpu().numpy().reshape(-1)
    a_o, c_o, d_o, _ = run_torso(sd, bg, enc_anchor, ind_t, None, 0.0, 0.8)
    bg_mixed = c_o * a_o + F32(1.0) * (F32(1) - a_o)
    assert np.array_equal(out["torso_color"].cpu().numpy(), bg_mixed)
    ref = render_inference(TriplaneSpec(1.0), sd, ro, rd, bits, enc_a, golden["net_ind"], golden["net_eye"], max_steps=48, bg_color=1.0)
    final = np.clip(ref["image_raw"] + (F32(1) - ref["weights_sum"])[:, None] * bg_mixed, F32(0), F32(1))
    assert np.array_equal(out["image"].cpu().numpy(), final)


def test_update_density_grid_torso_matches_checker():
    """torso half of update_extra_state (renderer.py:772-808): jittered points, forward_torso alpha, 5x5 max pool, EMA -- bit for bit"""
    from lzzx_nerf_amd.occupancy import update_density_grid_torso
    from lzzx_nerf_amd.torso import FusedTorso
    from oracle.occupancy import update_density_grid_torso as oracle_update
    sd = _torso_state(8)
    rng = np.random.default_rng(12)
    G = 64
    grid0 = rng.uniform(0, 0.6, G * G).astype(F32)
    noise = rng.uniform(0, 1, (G * G, 2)).astype(F32)
    ind = (rng.normal(size=(1, 8)) * 0.1).astype(F32)
    pose = np.eye(4, dtype=F32)
    pose[:3, 3] = [0.0, 0.1, 3.2]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torso = FusedTorso({k: torch.from_numpy(v) for k, v in sd.items()})
    enc_anchor = torso.encode_anchor(dev(pose[None]))
    dg = dev(grid0.copy())
    mean_g, thr_g = update_density_grid_torso(torso, dg, None, dev(ind), noise=dev(noise), enc_anchor=enc_anchor, density_thresh=0.01)
    ref = grid0.copy()
    mean_o = oracle_update(sd, ref, enc_anchor.cpu().numpy().reshape(-1), ind, noise)
    assert np.array_equal(dg.cpu().numpy(), ref)
    assert float(mean_g) == pytest.approx(mean_o, rel=2e-6) and float(thr_g) == pytest.approx(min(mean_o, 0.01), rel=2e-6)
    assert (ref != grid0 * F32(0.95)).mean() > 0.2     # the new alphas win somewhere, the decayed old grid elsewhere


def test_pipeline_smooth_lips_blends_audio_code_across_frames(params, golden):
    """opt.smooth_lips (renderer.py:254-258, 456-460): enc_a of frame k is 0.35 * (the blended code of frame k - 1) + 0.65 * its own,
    stateful across frames, and the frame is rendered with the blended code; bit for bit against the checker fed the same code"""
    from conftest import ellipsoid_bitfield, synthetic_camera
    from lzzx_nerf_amd.pipeline import TalkingHeadFrame, audio_window
    from oracle.audio import encode_audio
    from oracle.head import TriplaneSpec, get_rays
    from oracle.render import render_inference
    from test_audio_oracle import audio_state
    sd = dict(params)
    sd.update(audio_state(29, 32, True))
    H = W = 24
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = ellipsoid_bitfield()[0]
    track = np.random.default_rng(3).normal(size=(12, 29, 16)).astype(F32)          # a 12-frame audio feature track
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    frame = TalkingHeadFrame({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, dev(bits), bound=1.0, smooth_lips=True)
    prev = None
    for index in (0, 1, 2, 11):
        auds = audio_window(torch.from_numpy(track), 2, index)                      # att_mode 2: frames [index - 4, index + 4), zero-padded
        assert auds.shape == (8, 29, 16)
        out = frame.render(dev(ro), dev(rd), auds.cuda(), eye=dev(golden["net_eye"]), ind_code=dev(golden["net_ind"]), max_steps=32)
        own = encode_audio(sd, auds.numpy(), True)
        want = own if prev is None else F32(0.35) * prev + F32(1 - 0.35) * own
        assert np.array_equal(out["enc_a"].cpu().numpy(), want)
        ref = render_inference(TriplaneSpec(1.0), sd, ro, rd, bits, want, golden["net_ind"], golden["net_eye"], max_steps=32)
        assert np.array_equal(out["image"].cpu().numpy(), ref["image"])
        prev = want
    frame.reset()
    out = frame.render(dev(ro), dev(rd), auds.cuda(), eye=dev(golden["net_eye"]), ind_code=dev(golden["net_ind"]), max_steps=32)
    assert np.array_equal(out["enc_a"].cpu().numpy(), own)                          # a new clip starts from its own code


def test_torso_trains_through_the_operator_path():
    """forward_torso as an autograd graph over the operators (lzzx_nerf_amd.torso_train.TorsoTrainNet): the forward agrees with the
    one-launch inference kernel on the same state dict, and every gradient (MLP weights, tiled-grid table, anchor points) with a
    float64 torch model of the same math -- the reference trains exactly this graph (network.py:170-205, get_params :318-330)"""
    from lzzx_nerf_amd.torso import FusedTorso
    from lzzx_nerf_amd.torso_train import TorsoTrainNet
    sd = _torso_state(8, seed=3)
    net = TorsoTrainNet(ind_dim_torso=8).cuda()
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"torso_encoder.offsets"} or not missing.missing_keys
    rng = np.random.default_rng(9)
    N = 3000
    x = torch.from_numpy(rng.uniform(-1, 1, (N, 2)).astype(F32)).cuda()
    pose = np.eye(4, dtype=F32)
    pose[:3, 3] = [0.05, -0.02, 3.3]
    poses = torch.from_numpy(pose[None]).cuda()
    c = torch.from_numpy((rng.normal(size=(1, 8)) * 0.1).astype(F32)).cuda()
    target = torch.from_numpy(rng.uniform(0, 1, (N, 4)).astype(F32)).cuda()
    alpha, color, dx = net(x, poses, c)
    fused = FusedTorso({k: torch.from_numpy(v) for k, v in sd.items()})
    a_f, c_f, d_f = fused(x, poses, c)
    assert float((alpha - a_f).abs().max()) < 2e-5 and float((color - c_f).abs().max()) < 2e-5 and float((dx - d_f).abs().max()) < 2e-5
    loss = ((torch.cat([alpha, color], -1) - target) ** 2).mean() + 1e-2 * (dx ** 2).mean()
    loss.backward()
    grads = {k: p.grad.detach().double().cpu() for k, p in net.named_parameters() if p.grad is not None}
    assert {"torso_net.net.0.weight", "torso_deform_net.net.2.weight", "torso_encoder.embeddings", "anchor_points"} <= set(grads)

    # float64 model of the same graph in plain torch (dense bilinear interpolation of the tiled grid written out)
    P = {k: torch.from_numpy(v).double().requires_grad_(v.dtype == F32) for k, v in sd.items() if k != "torso_encoder.offsets"}
    offs = sd["torso_encoder.offsets"].astype(np.int64)

    def freq(v, deg):
        outs = [v]
        for k in range(deg):
            outs += [torch.sin(v * 2.0 ** k), torch.cos(v * 2.0 ** k)]
        return torch.cat(outs, -1)

    def tiled_grid(u01):
        feats = []
        S = np.log2(2048 / 16) / 15
        for l in range(16):
            scale = float(np.float32(np.exp2(np.float32(l) * np.float32(S)) * np.float32(16) - np.float32(1)))
            res = int(np.ceil(scale)) + 1
            size = int(offs[l + 1] - offs[l])
            pos = u01 * scale + 0.5
            g0 = torch.floor(pos).detach()
            fr = pos - g0
            g0 = g0.long()
            acc = 0
            for cx in (0, 1):
                for cy in (0, 1):
                    ix, iy = g0[:, 0] + cx, g0[:, 1] + cy
                    idx = (ix + iy * (res + 1)) % size                     # tiled: wrap the dense index (gridencoder.cu:54-72)
                    w = (fr[:, 0] if cx else 1 - fr[:, 0]) * (fr[:, 1] if cy else 1 - fr[:, 1])
                    acc = acc + w[:, None] * P["torso_encoder.embeddings"][offs[l] + idx]
            feats.append(acc)
        return torch.cat(feats, -1)

    xd = x.double().cpu() * 0.8
    wrapped = P["anchor_points"][None] @ torch.from_numpy(pose[None]).double().permute(0, 2, 1).inverse()
    wrapped = (wrapped[:, :, :2] / wrapped[:, :, 3, None] / wrapped[:, :, 2, None]).view(1, -1)
    h = torch.cat([freq(xd, 8), freq(wrapped, 3).repeat(N, 1), c.double().cpu().repeat(N, 1)], -1)
    mlp = lambda v, name: torch.relu(torch.relu(v @ P[f"{name}.net.0.weight"].T) @ P[f"{name}.net.1.weight"].T) @ P[f"{name}.net.2.weight"].T
    dxd = mlp(h, "torso_deform_net")
    xx = (xd + dxd).clamp(-1, 1)
    hh = mlp(torch.cat([tiled_grid((xx + 1) / 2), h], -1), "torso_net")
    out = torch.sigmoid(hh) * 1.002 - 0.001
    ref_loss = ((out - target.double().cpu()) ** 2).mean() + 1e-2 * (dxd ** 2).mean()
    ref_loss.backward()
    assert float(loss) == pytest.approx(float(ref_loss), rel=1e-5)
    for k, g in grads.items():
        r = P[k].grad
        scale = float(r.abs().max())
        assert scale > 0 and float((g - r).abs().max()) <= 2e-3 * scale, (k, float((g - r).abs().max()), scale)
