#!/usr/bin/env python3
"""Generate tests/golden/reference_frontends.npz from the REFERENCE's own Python (build container only; see make_golden.py for the
back-end adapters: the four CUDA extensions are replaced by adapters onto the CPU checker, so what the vectors pin is the reference's
*Python-level* arithmetic -- torch conv1d / linear / softmax / grid_sample / max_pool2d / matmul / boolean-mask EMA -- of

  * nerf_triplane.utils.get_rays (batched poses, N > 0 random, patch_size > 1, rect) and get_bg_coords        utils.py:217-312
  * nerf_triplane.network.AudioNet / AudioAttNet / NeRFNetwork.encode_audio (29, 44 and 1024 input channels)  network.py:9-70,226-240
  * NeRFNetwork.forward_torso and NeRFRenderer.run_torso (opt.torso = True)                                   network.py:170-205, renderer.py:572-631
  * NeRFRenderer.mark_untrained_grid and update_extra_state, head and torso branches, on a 16^3 / 32^2 grid    renderer.py:633-818

Large inputs and weights are NOT stored: they are drawn from numpy generators with the seeds below and the tests regenerate them
(`frontends_inputs`), so the fixture holds outputs only (plus torch-RNG draws the reference makes internally: pixel indices, jitter).

Run:  python tests/golden/make_golden_frontends.py
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs sys.path for the reference and the checker)
from make_golden import O, _t2n  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from frontends_inputs import (GRID, GRID3, audio_weights, audio_windows, camera_set, head_pose, initial_density_grid,  # noqa: E402
                              torso_pixels, torso_weights)


# ---- the generator -----------------------------------------------------------------------------------------------------------------
class RandRecorder:
    """records what torch.rand_like returns while the reference runs (update_extra_state's jitter, renderer.py:751,796)"""

    def __init__(self):
        self.draws, self._orig = [], torch.rand_like

    def __enter__(self):
        def rec(t, *a, **k):
            r = self._orig(t, *a, **k)
            self.draws.append(r.clone().numpy())
            return r
        torch.rand_like = rec
        return self

    def __exit__(self, *exc):
        torch.rand_like = self._orig


def install_raymarching_backend():
    be = sys.modules["_raymarching_face"]
    be.morton3D = lambda coords, N, indices: indices.copy_(torch.from_numpy(O.morton3D(_t2n(coords))))
    be.morton3D_invert = lambda indices, N, coords: coords.copy_(torch.from_numpy(O.morton3D_invert(_t2n(indices))))
    be.packbits = lambda grid, N, thresh, bitfield: bitfield.copy_(torch.from_numpy(O.packbits(_t2n(grid), float(thresh))))
    be.morton3D_dilation = lambda grid, C, H, out: out.copy_(torch.from_numpy(O.morton3D_dilation(_t2n(grid))))
    torch.Tensor.cuda = lambda self, *a, **k: self   # the raymarching wrappers move CPU inputs with .cuda() (raymarching.py:94); no GPU here


def load_np(net, P):
    sd = net.state_dict()
    for k, v in P.items():
        if k in sd:
            assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(v))


def shrink_grids(net, G):
    net.grid_size = G
    net.density_grid = torch.zeros(net.cascade, G ** 3)
    net.density_bitfield = torch.zeros(net.cascade * G ** 3 // 8, dtype=torch.uint8)
    if net.torso:
        net.density_grid_torso = torch.zeros(G ** 2)


def main():
    MG.install_backends()
    install_raymarching_backend()
    from nerf_triplane.network import NeRFNetwork  # reference
    from nerf_triplane.utils import get_bg_coords, get_rays  # reference
    out = {}
    t = torch.from_numpy

    # ---- (1) get_rays / get_bg_coords -------------------------------------------------------------------------------------------
    H, W = 48, 64
    intr = [130.0, 125.0, 31.0, 25.5]
    poses = np.stack([head_pose(0.0, 0.0), head_pose(0.3, 0.1)])
    out["rays_poses"], out["rays_intr"], out["rays_HW"] = poses, np.array(intr), np.array([H, W])
    for tag, kw in (("full", dict(N=-1)), ("rand", dict(N=300)), ("patch", dict(N=256, patch_size=4)), ("rect", dict(rect=(5, 20, 3, 40))),
                    ("clamp", dict(N=10 ** 6))):
        torch.manual_seed(7)
        r = get_rays(t(poses if tag != "rect" else poses[:1]), intr, H, W, **kw)
        for k in ("i", "j", "inds", "rays_o", "rays_d"):
            if tag == "clamp" and k != "inds":
                continue    # N > H*W clamps to H*W random pixels (utils.py:250): the index draw is the point
            out[f"rays_{tag}_{k}"] = r[k].contiguous().numpy()
    out["bg_coords_48_64"] = get_bg_coords(H, W, "cpu").numpy()
    out["bg_coords_5_7"] = get_bg_coords(5, 7, "cpu").numpy()

    # ---- (2) audio front-end ------------------------------------------------------------------------------------------------------
    for asr, dim_in in (("deepspeech", 29), ("esperanto", 44), ("hubert", 1024)):
        opt = MG.Opt()
        opt.asr_model = asr
        net = NeRFNetwork(opt).eval()
        assert net.audio_in_dim == dim_in
        load_np(net, audio_weights(dim_in))
        a = t(audio_windows(dim_in))
        with torch.no_grad():
            out[f"audio_{dim_in}_feat"] = net.audio_net(a).numpy()          # AudioNet alone, [8, 32]
            out[f"audio_{dim_in}_enc_a"] = net.encode_audio(a).numpy()      # + AudioAttNet, [1, 32]
            net.att = 0
            out[f"audio_{dim_in}_noatt"] = net.encode_audio(a[:1]).numpy()  # opt.att = 0: one window, no attention

    # ---- (3) torso: forward_torso, run_torso, torso half of update_extra_state ----------------------------------------------------
    opt = MG.Opt()
    opt.torso = True
    torch.manual_seed(0)
    net = NeRFNetwork(opt).eval()
    TP = torso_weights()
    load_np(net, TP)
    shrink_grids(net, GRID)
    pose = head_pose()
    x = torso_pixels()
    with torch.no_grad():
        alpha, color, dx = net.forward_torso(t(x), t(pose[None]), t(TP["individual_codes_torso"][:1]))
        out.update(torso_alpha=alpha.numpy(), torso_color=color.numpy(), torso_dx=dx.numpy())
        # anchor encoding alone (what FusedTorso.encode_anchor restates)
        wa = net.anchor_points[None, ...] @ t(pose[None]).permute(0, 2, 1).inverse()
        wa = (wa[:, :, :2] / wa[:, :, 3, None] / wa[:, :, 2, None]).view(1, -1)
        out["torso_enc_anchor"] = net.anchor_encoder(wa).numpy()
        # run_torso: 2-D occupancy mask from a random grid, masked query, mix with a white background
        rng = np.random.default_rng(17)
        net.density_grid_torso.copy_(t(rng.uniform(0, 0.02, GRID ** 2).astype(np.float32)))
        net.mean_density_torso = 0.012          # -> threshold min(0.01, 0.012) = 0.01 (renderer.py:603)
        bg = get_bg_coords(24, 20, "cpu")
        res = net.run_torso(torch.zeros(1, 24 * 20, 3), bg, t(pose[None]), index=0, bg_color=1)
        out.update(run_torso_grid=net.density_grid_torso.numpy().copy(), run_torso_alpha=res["torso_alpha"].numpy(),
                   run_torso_bg=res["bg_color"].numpy(), run_torso_hw=np.array([24, 20]))
        # torso half of update_extra_state
        net.poses = t(pose[None])
        net.aud_features = t(audio_windows(29))
        net.eye_area = torch.full((8, 1), 0.25)
        load_np(net, audio_weights(29))
        net.individual_codes_torso.data[0].copy_(t(TP["individual_codes_torso"][0]))
        net.density_grid_torso.copy_(t(rng.uniform(0, 1, GRID ** 2).astype(np.float32)))
        out["occ_torso_grid0"] = net.density_grid_torso.numpy().copy()
        random.seed(0)
        torch.manual_seed(21)
        with RandRecorder() as rec:
            net.update_extra_state()
        assert len(rec.draws) == 1 and rec.draws[0].shape == (GRID ** 2, 2)
        out.update(occ_torso_noise=rec.draws[0], occ_torso_grid1=net.density_grid_torso.numpy().copy(),
                   occ_torso_mean=np.array([net.mean_density_torso], np.float64))

    # ---- (4) head occupancy grid: mark_untrained_grid + two update_extra_state calls ------------------------------------------------
    golden = np.load(os.path.join(HERE, "reference_python.npz"))
    for bound in (1, 2):
        opt = MG.Opt()
        opt.bound = bound
        torch.manual_seed(0)
        net = NeRFNetwork(opt).eval()
        shrink_grids(net, GRID3)
        # head weights: the committed fixture state-dict (bound 1) or seeded ones; tables from the usual seed
        sd = {k[3:]: golden[k] for k in golden.files if k.startswith("sd/") and ".offsets" not in k}
        load_np(net, sd)
        trng = np.random.default_rng(1234 + bound)
        for n in ("xy", "yz", "xz"):
            enc = getattr(net, f"encoder_{n}")
            enc.embeddings.data.copy_(t(trng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float32)))
        load_np(net, audio_weights(29))
        net.aud_features = t(audio_windows(29))
        net.eye_area = torch.full((8, 1), 0.25)
        net.density_grid.copy_(t(initial_density_grid(net.cascade)))
        cams = camera_set()
        cam_intr = [2.667 * 64, 2.667 * 64, 32.0, 32.0]
        net.mark_untrained_grid(cams, cam_intr)
        out[f"occ_b{bound}_marked"] = net.density_grid.numpy().copy()
        out[f"occ_b{bound}_cams"], out[f"occ_b{bound}_cam_intr"] = cams, np.array(cam_intr)
        for it in range(2):
            random.seed(3 + it)
            torch.manual_seed(30 + it)
            with RandRecorder() as rec:
                net.update_extra_state()
            assert len(rec.draws) == net.cascade
            rand_idx = random.Random(3 + it).randint(0, 7)
            with torch.no_grad():
                from nerf_triplane.utils import get_audio_features
                enc_a = net.encode_audio(get_audio_features(net.aud_features, net.att, rand_idx))
            out[f"occ_b{bound}_it{it}_noise"] = np.stack(rec.draws)
            out[f"occ_b{bound}_it{it}_enc_a"] = enc_a.numpy()
            out[f"occ_b{bound}_it{it}_grid"] = net.density_grid.numpy().copy()
            out[f"occ_b{bound}_it{it}_bits"] = net.density_bitfield.numpy().copy()
            out[f"occ_b{bound}_it{it}_mean"] = np.array([net.mean_density], np.float64)
    path = os.path.join(HERE, "reference_frontends.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB;", len(out), "arrays")


if __name__ == "__main__":
    main()
