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


from lzzx_nerf_amd.synthetic import ellipsoid_grid, make_params, synthetic_camera  # noqa: E402,F401  (shared with bench.py / smoke())


@pytest.fixture(scope="session")
def params(golden):
    return make_params(golden)


def ellipsoid_bitfield(grid_size=128, semi=(0.35, 0.45, 0.35), bound=1.0):
    """SURVEY 8d occupancy variant (i) through the CHECKER's morton3D / packbits (CPU tests have no GPU; the product builds the same
    bitfield with its HIP operators: lzzx_nerf_amd.synthetic.ellipsoid_bitfield_device, compared in test_gpu_parity)"""
    from oracle import oracle as O
    inside, coords = ellipsoid_grid(grid_size, semi, bound)
    idx = O.morton3D(coords)
    grid = np.zeros((1, grid_size ** 3), dtype=np.float32)
    grid[0, idx] = inside.astype(np.float32)
    return O.packbits(grid, 0.5), grid
