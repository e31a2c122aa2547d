import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """make sure the CPU checker and the HIP library exist (both are built by __graft_entry__.build();
    hipcc cross-compiles without a GPU).  Building the checker here is test infrastructure."""
    from oracle import oracle as O
    O.build()
    from lzzx_nerf_amd import build as B
    if not os.path.exists(B.SO):
        B.build()
    yield


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"), allow_pickle=False)


def make_params(golden, seed_tables=1234):
    """state-dict-shaped numpy weights: MLP weights from the reference fixture, tables regenerated from the seed
    used by tests/golden/make_golden.py"""
    P = {k[3:]: golden[k] for k in golden.files if k.startswith("sd/")}
    rng = np.random.default_rng(seed_tables)
    n = int(P["encoder_xy.offsets"][-1])
    for name in ("xy", "yz", "xz"):
        P[f"encoder_{name}.embeddings"] = rng.uniform(-1, 1, (n, 1)).astype(np.float32)
    return P


@pytest.fixture(scope="session")
def params(golden):
    return make_params(golden)


def synthetic_camera(H, W):
    """SURVEY 8d synthetic camera: identity rotation, t = (0, 0, -3.35), fovy 21.24 deg"""
    fl = H / (2 * np.tan(np.radians(21.24) / 2))
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = -3.35
    return pose, [fl, fl, W / 2, H / 2]


def ellipsoid_bitfield(grid_size=128, semi=(0.35, 0.45, 0.35), bound=1.0):
    """SURVEY 8d occupancy variant (i): head-like ellipsoid, Morton-ordered, packed"""
    from oracle import oracle as O
    c = np.arange(grid_size, dtype=np.int32)
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    xyz = (coords.astype(np.float32) + 0.5) / grid_size * 2 * bound - bound
    inside = ((xyz / np.array(semi, dtype=np.float32)) ** 2).sum(1) <= 1.0
    idx = O.morton3D(coords)
    grid = np.zeros((1, grid_size ** 3), dtype=np.float32)
    grid[0, idx] = inside.astype(np.float32)
    return O.packbits(grid, 0.5), grid
