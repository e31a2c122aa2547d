#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own pure-Python code (run in the build container only;
/root/reference does not exist on the GPU box and nothing at test time imports it).

What is imported from /root/reference and what it pins:
  * encoding.get_encoder / gridencoder.grid.GridEncoder.__init__   -> table layout (offsets, shapes, output_dim)
  * nerf_triplane.network.NeRFNetwork (constructed on CPU)          -> state_dict keys/shapes, MLP / density /
    forward arithmetic (torch CPU, fp32) on seeded weights and seeded inputs
  * nerf_triplane.utils.get_rays                                    -> ray generation
The four compiled CUDA back-ends (`_gridencoder`, `_shencoder`, `_freqencoder`, `_raymarching_face`) cannot be
built here (nvcc absent).  They are replaced in sys.modules by thin adapters that forward the reference
wrappers' calls to the CPU checker (oracle/), so that the reference's *Python* (GridEncoder.forward bookkeeping,
NeRFNetwork.forward / density) runs end to end; absent third-party packages the hot path never touches
(lpips, trimesh, mcubes, ...) are replaced by empty modules.  The vectors therefore pin the reference's
Python-level semantics (layouts, permutes, concatenation order, activations), not its CUDA arithmetic.

Run:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import oracle as O  # noqa: E402


def _t2n(t):
    return t.detach().cpu().numpy()


def install_backends():
    ge = types.ModuleType("_gridencoder")

    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners):
        out, dd = O.grid_encode_forward(_t2n(inputs), _t2n(embeddings), _t2n(offsets), float(2.0 ** np.float64(S)), H,
                                        dy_dx is not None, gridtype, align_corners)
        # wrapper expects level-major [L, B, C]
        outputs.copy_(torch.from_numpy(out.reshape(B, L, C).transpose(1, 0, 2).copy()))
        if dy_dx is not None:
            dy_dx.copy_(torch.from_numpy(dd))

    ge.grid_encode_forward = grid_encode_forward
    ge.grid_encode_backward = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    sys.modules["_gridencoder"] = ge

    sh = types.ModuleType("_shencoder")

    def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
        out, dd = O.sh_encode_forward(_t2n(inputs), C, dy_dx is not None)
        outputs.copy_(torch.from_numpy(out))
        if dy_dx is not None:
            dy_dx.copy_(torch.from_numpy(dd))

    sh.sh_encode_forward = sh_encode_forward
    sys.modules["_shencoder"] = sh

    fr = types.ModuleType("_freqencoder")

    def freq_encode_forward(inputs, B, D, deg, C, outputs):
        outputs.copy_(torch.from_numpy(O.freq_encode_forward(_t2n(inputs), deg)))

    fr.freq_encode_forward = freq_encode_forward
    sys.modules["_freqencoder"] = fr
    sys.modules["_raymarching_face"] = types.ModuleType("_raymarching_face")
    for name in ("lpips", "trimesh", "mcubes", "cv2", "tensorboardX", "torch_ema", "imageio", "pydub", "numba", "tqdm.rich"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


class Opt:
    """hot-path defaults of train.py:32-107 / HubertInferenceMQ.py:25-96"""
    bound = 1
    min_near = 0.05
    density_thresh = 10
    density_thresh_torso = 0.01
    exp_eye = True
    test_train = False
    smooth_lips = False
    torso = False
    cuda_ray = True
    ind_num = 10
    ind_dim = 4
    ind_dim_torso = 8
    train_camera = False
    emb = False
    asr_model = "deepspeech"
    att = 2
    unc_loss = 1
    torso_shrink = 0.8


def main():
    install_backends()
    import encoding  # reference
    from nerf_triplane.network import NeRFNetwork  # reference
    from nerf_triplane.utils import get_rays  # reference

    out = {}
    # ---- (1) table layouts -------------------------------------------------------------------------
    for tag, kw in (("triplane", dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=14,
                                      desired_resolution=512)),
                    ("hashgrid_default", dict()),
                    ("tiled_torso", dict(input_dim=2, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                                         desired_resolution=2048))):
        enc, od = encoding.get_encoder("tiledgrid" if tag == "tiled_torso" else "hashgrid", **kw)
        out[f"layout_{tag}_offsets"] = enc.offsets.numpy()
        out[f"layout_{tag}_shape"] = np.array(enc.embeddings.shape)
        out[f"layout_{tag}_outdim"] = np.array([od])
        out[f"layout_{tag}_pls"] = np.array([enc.per_level_scale], dtype=np.float64)
    # ---- (2) network ---------------------------------------------------------------------------------
    torch.manual_seed(0)
    net = NeRFNetwork(Opt())
    net.eval()
    # tables with a numerically meaningful range
    # (regenerated from the seed by the tests instead of being stored: 3 x 163584 floats)
    trng = np.random.default_rng(1234)
    for n in ("xy", "yz", "xz"):
        enc = getattr(net, f"encoder_{n}")
        enc.embeddings.data.copy_(torch.from_numpy(trng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float32)))
    sd = net.state_dict()
    keep = [k for k in sd if k.startswith(("sigma_net", "color_net", "unc_net", "aud_ch_att_net", "eye_att_net"))
            or k.endswith(".offsets") or k == "individual_codes"]
    for k in keep:
        out["sd/" + k] = sd[k].numpy()
    out["sd_keys"] = np.array(sorted(sd.keys()))
    out["sd_shapes"] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])

    M = 777
    rng = np.random.default_rng(2)
    xyz = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    xyz[0] = [1, 1, 1]
    xyz[1] = [-1, -1, -1]
    xyz[2] = [0, 0, 0]
    xyz[3] = [1.5, 0.2, 0.1]  # out of range -> zero features on the planes that see x
    d = rng.normal(size=(M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    enc_a = rng.normal(size=(1, 32)).astype(np.float32)
    eye = np.array([[0.25]], dtype=np.float32)
    ind = sd["individual_codes"][0].numpy()
    with torch.no_grad():
        tx, td = torch.from_numpy(xyz), torch.from_numpy(d)
        enc_x = net.encode_x(tx, bound=net.bound)
        dres = net.density(tx, torch.from_numpy(enc_a), torch.from_numpy(eye), enc_x)
        net.testing = True
        sig, rgb, aa, ae, unc_test = net(tx, td, torch.from_numpy(enc_a), torch.from_numpy(ind), torch.from_numpy(eye))
        net.testing = False
        _, _, _, _, unc_train = net(tx, td, torch.from_numpy(enc_a), torch.from_numpy(ind), torch.from_numpy(eye))
        # per-layer activations of each MLP on the same features (golden for lzo_linear)
        h = enc_x
        a1 = torch.relu(net.aud_ch_att_net.net[0](h))
        att = net.aud_ch_att_net.net[1](a1)
    out.update(net_xyz=xyz, net_dirs=d, net_enc_a=enc_a, net_eye=eye, net_ind=ind, net_enc_x=enc_x.numpy(),
               net_sigma=sig.numpy(), net_rgb=rgb.numpy(), net_amb_aud=aa.numpy(), net_amb_eye=ae.numpy(),
               net_unc_test_first=unc_test.numpy().reshape(M, -1)[:, :1], net_unc_test_shape=np.array(unc_test.shape),
               net_unc_train=unc_train.numpy(), net_geo=dres["geo_feat"].numpy(), net_aud_hidden=a1.numpy(), net_att=att.numpy())
    # ---- (3) rays ------------------------------------------------------------------------------------
    for HW in (64, 256):
        fl = HW / (2 * np.tan(np.radians(21.24) / 2))
        pose = np.eye(4, dtype=np.float32)
        pose[2, 3] = -3.35
        # a second, rotated pose
        th = 0.3
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], dtype=np.float32)
        pose2 = np.eye(4, dtype=np.float32)
        pose2[:3, :3] = R
        pose2[:3, 3] = R @ np.array([0, 0, -3.35], dtype=np.float32)
        for tag, p in (("id", pose), ("rot", pose2)):
            r = get_rays(torch.from_numpy(p[None]), [fl, fl, HW / 2, HW / 2], HW, HW, -1)
            ro, rd = r["rays_o"][0].numpy(), r["rays_d"][0].numpy()
            if HW == 64:
                out[f"rays_{tag}_{HW}_o"] = ro
                out[f"rays_{tag}_{HW}_d"] = rd
            else:  # keep the fixture small: every 97th ray + a float64 checksum
                out[f"rays_{tag}_{HW}_d_sub"] = rd[::97]
                out[f"rays_{tag}_{HW}_d_sum"] = rd.astype(np.float64).sum(0)
            out[f"rays_{tag}_{HW}_pose"] = p
            out[f"rays_{tag}_{HW}_intr"] = np.array([fl, fl, HW / 2, HW / 2], dtype=np.float64)
    path = os.path.join(HERE, "reference_python.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
