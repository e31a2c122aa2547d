"""Seeded inputs of tests/golden/reference_frontends.npz: weights, audio windows, poses, pixels, initial grids.  Shared by the generator
(tests/golden/make_golden_frontends.py, which feeds them to the reference's Python in the build container) and by the tests (which feed
them to the checker / the HIP path and compare with the stored reference outputs).  numpy generators only: no reference code, no torch RNG."""
import numpy as np

from oracle import oracle as O

GRID = 32    # torso density grid of the fixture (the reference hard-codes 128, renderer.py:94; every formula uses self.grid_size)
GRID3 = 16   # head density grid of the fixture (the recorded jitter is 12 bytes per cell and does not compress)


def audio_weights(dim_in, dim_aud=32, seed=11):
    """state-dict-shaped AudioNet + AudioAttNet weights, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from a numpy generator"""
    rng = np.random.default_rng(seed + dim_in)
    u = lambda shape, fan: rng.uniform(-1, 1, shape).astype(np.float32) / np.float32(np.sqrt(fan))
    P = {}
    for idx, (ci, co) in zip((0, 2, 4, 6), ((dim_in, 32), (32, 32), (32, 64), (64, 64))):
        P[f"audio_net.encoder_conv.{idx}.weight"], P[f"audio_net.encoder_conv.{idx}.bias"] = u((co, ci, 3), 3 * ci), u((co,), 3 * ci)
    for idx, (ci, co) in zip((0, 2), ((64, 64), (64, dim_aud))):
        P[f"audio_net.encoder_fc1.{idx}.weight"], P[f"audio_net.encoder_fc1.{idx}.bias"] = u((co, ci), ci), u((co,), ci)
    for idx, (ci, co) in zip((0, 2, 4, 6, 8), ((dim_aud, 16), (16, 8), (8, 4), (4, 2), (2, 1))):
        P[f"audio_att_net.attentionConvNet.{idx}.weight"], P[f"audio_att_net.attentionConvNet.{idx}.bias"] = u((co, ci, 3), 3 * ci), u((co,), 3 * ci)
    P["audio_att_net.attentionNet.0.weight"], P["audio_att_net.attentionNet.0.bias"] = u((8, 8), 8), u((8,), 8)
    return P


def audio_windows(dim_in, seed=12):
    return np.random.default_rng(seed + dim_in).normal(size=(8, dim_in, 16)).astype(np.float32)


def torso_weights(seed=13):
    rng = np.random.default_rng(seed)
    lin = lambda n, k: (rng.uniform(-1, 1, (n, k)) / np.sqrt(k)).astype(np.float32)
    offs = O.grid_offsets(2, 16, np.exp2(np.log2(2048 / 16) / 15), 16, 16)
    return {"anchor_points": np.array([[0.01, 0.01, 0.1, 1], [-0.1, -0.1, 0.1, 1], [0.1, -0.1, 0.1, 1]], np.float32),
            "torso_deform_net.net.0.weight": lin(32, 84), "torso_deform_net.net.1.weight": lin(32, 32),
            "torso_deform_net.net.2.weight": lin(2, 32), "torso_net.net.0.weight": lin(32, 116), "torso_net.net.1.weight": lin(32, 32),
            "torso_net.net.2.weight": lin(4, 32), "torso_encoder.offsets": offs.astype(np.int32),
            "torso_encoder.embeddings": rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32),
            "individual_codes_torso": (rng.normal(size=(10, 8)) * 0.1).astype(np.float32)}


def head_pose(th=0.2, tx=0.05):
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], dtype=np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R
    pose[:3, 3] = R @ np.array([tx, 0, -3.35], dtype=np.float32)
    return pose


def torso_pixels(n=600, seed=14):
    x = np.random.default_rng(seed).uniform(-1, 1, (n, 2)).astype(np.float32)
    x[0], x[1], x[2] = [-1, -1], [1, 1], [0, 0]
    return x


def camera_set(n=5, seed=15):
    rng = np.random.default_rng(seed)
    return np.stack([head_pose(th, tx) for th, tx in zip(rng.uniform(-0.4, 0.4, n), rng.uniform(-0.2, 0.2, n))])


def initial_density_grid(cascade, seed=16):
    rng = np.random.default_rng(seed)
    g = rng.uniform(0, 2, (cascade, GRID3 ** 3)).astype(np.float32)
    g[rng.uniform(size=g.shape) < 0.2] = 0.0
    return g
