"""Occupancy-grid maintenance: lzzx_nerf_amd.occupancy.update_density_grid against the CPU restatement of
NeRFRenderer.update_extra_state's head branch (renderer.py:699-766) on the same noise: query points, densities, dilated EMA grid
and bitfield bit for bit; the mean (a free-order f32 sum on both sides) to rounding."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle.head import TriplaneSpec
from oracle.occupancy import update_density_grid as oracle_update

pytestmark = pytest.mark.gpu


def _params_for_bound(params, bound, rng):
    if bound == 1.0:
        return params
    spec = TriplaneSpec(bound)
    p = dict(params)
    for n in ("xy", "yz", "xz"):
        p[f"encoder_{n}.embeddings"] = rng.uniform(-1, 1, (spec.n_params, 1)).astype(np.float32)
        p[f"encoder_{n}.offsets"] = spec.offsets.astype(np.int32)
    return p


@pytest.mark.parametrize("bound,G,thresh", [(1.0, 32, 0.01), (2.0, 16, 0.01), (1.0, 16, 1e9)])
def test_update_density_grid_matches_checker(params, golden, bound, G, thresh):
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.occupancy import update_density_grid
    rng = np.random.default_rng(int(bound * 10) + G)
    p = _params_for_bound(params, bound, rng)
    C = 1 + int(np.ceil(np.log2(bound)))
    cells = G ** 3
    grid0 = rng.uniform(0, 2, (C, cells)).astype(np.float32)
    grid0[rng.uniform(size=grid0.shape) < 0.2] = -1.0      # untrained cells stay untouched (renderer.py:763)
    grid0[rng.uniform(size=grid0.shape) < 0.2] = 0.0
    noise = rng.uniform(0, 1, (C, cells, 3)).astype(np.float32)
    enc_a, eye = golden["net_enc_a"], golden["net_eye"]
    ref_grid = grid0.copy()
    mean_o, thresh_o, bits_o = oracle_update(TriplaneSpec(bound), p, ref_grid, enc_a, eye, bound, noise, density_thresh=thresh)
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in p.items()}, bound=bound)
    dg = torch.from_numpy(grid0.copy()).cuda()
    bf = torch.zeros(C * cells // 8, dtype=torch.uint8, device="cuda")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    mean_g, thresh_g = update_density_grid(head, dg, bf, dev(enc_a), dev(eye), bound=bound, density_thresh=thresh, noise=dev(noise))
    assert np.array_equal(dg.cpu().numpy(), ref_grid)                       # dilated EMA grid: bit for bit
    assert float(mean_g) == pytest.approx(mean_o, rel=2e-6)
    if thresh < 1:   # threshold = density_thresh exactly on both sides
        assert float(thresh_g) == np.float32(thresh) and np.array_equal(bf.cpu().numpy(), bits_o)
    else:            # threshold = mean: equal up to the cells whose value is within rounding of the mean
        near = np.abs(ref_grid - mean_o) < 1e-5 * max(mean_o, 1e-6)
        mism = np.unpackbits(bf.cpu().numpy() ^ bits_o, bitorder="little").astype(bool)
        assert not (mism & ~near.reshape(-1)).any()
    assert float((dg >= 0).float().mean()) > 0.5 and int((dg == -1).sum()) == int((grid0 == -1).sum())


def test_update_density_grid_feeds_the_march(params, golden):
    """the bitfield it writes is what march_rays consumes: a render with it is well-formed and skips cells"""
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.occupancy import update_density_grid
    from lzzx_nerf_amd.renderer import TriplaneRenderer, get_rays
    from conftest import synthetic_camera
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dg = torch.zeros(1, 128 ** 3, device="cuda")
    bf = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device="cuda")
    torch.manual_seed(0)
    enc_a, eye, ind = dev(golden["net_enc_a"]), dev(golden["net_eye"]), dev(golden["net_ind"])
    # random-init weights give sigma ~ 1 everywhere: a large density_thresh makes the threshold the mean density (renderer.py:770)
    mean, thresh = update_density_grid(head, dg, bf, enc_a, eye, bound=1.0, density_thresh=10.0)
    occ = float(np.unpackbits(bf.cpu().numpy()).mean())
    assert 0.0 < occ < 1.0 and float(mean) > 0 and float(thresh) == float(mean)
    pose, intr = synthetic_camera(64, 64)
    ro, rd = get_rays(dev(pose), intr, 64, 64)
    out = TriplaneRenderer(head, bf, bound=1.0).render(ro, rd, enc_a, ind, eye, max_steps=64)
    assert bool(torch.isfinite(out["image"]).all()) and int(out["state"][3]) == 1
