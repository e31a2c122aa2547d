"""The checker's restatements of the path's neighbours -- ray selection (utils.py:217-312), the audio front-end (network.py:9-70,
226-240), the torso branch (network.py:170-205, renderer.py:572-631) and the occupancy-grid maintenance (renderer.py:633-818) --
against outputs of the REFERENCE's own Python, generated in the build container by tests/golden/make_golden_frontends.py on the
seeded inputs of tests/frontends_inputs.py (the CUDA back-ends under that Python were adapters onto the checker's C kernels, so these
vectors pin the torch-level arithmetic and bookkeeping: conv1d / linear / softmax / grid_sample / max_pool2d / matmul / masked EMA /
index rules).  The GPU path is compared with the same vectors in tests/test_gpu_frontends.py."""
import os

import numpy as np
import pytest
import torch

import frontends_inputs as FI
from oracle import oracle as O
from oracle.head import TriplaneSpec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fe():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_frontends.npz"), allow_pickle=False)


# ---- get_rays / get_bg_coords ---------------------------------------------------------------------------------------------------
RAY_CASES = [("full", dict(N=-1)), ("rand", dict(N=300)), ("patch", dict(N=256, patch_size=4)), ("rect", dict(rect=(5, 20, 3, 40)))]


@pytest.mark.parametrize("tag,kw", RAY_CASES)
def test_get_rays_checker_matches_reference(fe, tag, kw):
    H, W = [int(v) for v in fe["rays_HW"]]
    poses = fe["rays_poses"] if tag != "rect" else fe["rays_poses"][:1]
    inds = fe[f"rays_{tag}_inds"]
    assert all(np.array_equal(inds[0], row) for row in inds)            # one index list shared by the batch (expand)
    r = O.get_rays_batched(poses, fe["rays_intr"], H, W, None if tag == "full" else inds[0])
    assert np.array_equal(r["i"], fe[f"rays_{tag}_i"]) and np.array_equal(r["j"], fe[f"rays_{tag}_j"])
    assert np.array_equal(r["rays_o"], fe[f"rays_{tag}_rays_o"])
    assert np.max(np.abs(r["rays_d"] - fe[f"rays_{tag}_rays_d"])) < 2e-7   # torch.matmul fixes no summation order


@pytest.mark.parametrize("tag,kw", RAY_CASES[1:] + [("clamp", dict(N=10 ** 6))])
def test_pixel_selection_draws_like_the_reference(fe, tag, kw):
    """the host-side index rules of lzzx_nerf_amd.utils (same torch generator calls in the same order as utils.py:252-285)"""
    from lzzx_nerf_amd.utils import select_pixels
    H, W = [int(v) for v in fe["rays_HW"]]
    torch.manual_seed(7)
    sel = select_pixels(H, W, device="cpu", **kw)
    assert sel.dtype == torch.int64 and np.array_equal(sel.numpy(), fe[f"rays_{tag}_inds"][0])
    assert select_pixels(H, W, -1, device="cpu") is None


def test_bg_coords_checker_matches_reference(fe):
    assert np.array_equal(O.bg_coords(48, 64), fe["bg_coords_48_64"][0])
    assert np.array_equal(O.bg_coords(5, 7), fe["bg_coords_5_7"][0])


# ---- audio front-end ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim_in", [29, 44, 1024])
def test_audio_checker_matches_reference(fe, dim_in):
    from oracle.audio import encode_audio
    P, a = FI.audio_weights(dim_in), FI.audio_windows(dim_in)
    feat = encode_audio(P, a, use_att=False)
    scale = np.abs(fe[f"audio_{dim_in}_feat"]).max()
    assert np.max(np.abs(feat - fe[f"audio_{dim_in}_feat"])) < 3e-6 * max(scale, 1)        # AudioNet (cuDNN / MKL fix no order)
    assert np.max(np.abs(encode_audio(P, a) - fe[f"audio_{dim_in}_enc_a"])) < 3e-6 * max(scale, 1)   # + AudioAttNet
    assert np.max(np.abs(encode_audio(P, a[:1], use_att=False) - fe[f"audio_{dim_in}_noatt"])) < 3e-6 * max(scale, 1)


# ---- torso ------------------------------------------------------------------------------------------------------------------------
def test_torso_checker_matches_reference(fe):
    from oracle.torso import encode_anchor, forward_torso, run_torso
    P = FI.torso_weights()
    pose = FI.head_pose()
    ea = encode_anchor(P, pose)
    assert np.max(np.abs(ea - fe["torso_enc_anchor"])) < 1e-5
    alpha, color, dx = forward_torso(P, FI.torso_pixels(), fe["torso_enc_anchor"], P["individual_codes_torso"][0])
    assert np.max(np.abs(dx - fe["torso_dx"])) < 2e-6
    assert np.max(np.abs(alpha - fe["torso_alpha"])) < 2e-6 and np.max(np.abs(color - fe["torso_color"])) < 2e-6
    # run_torso: occupancy mask -> masked query -> zeros elsewhere -> mixed with the white background (renderer.py:603-621)
    Hh, Ww = [int(v) for v in fe["run_torso_hw"]]
    a, c, _, mask = run_torso(P, O.bg_coords(Hh, Ww), fe["torso_enc_anchor"], P["individual_codes_torso"][0], fe["run_torso_grid"], 0.01)
    assert np.array_equal(mask, fe["run_torso_alpha"][:, 0] != 0)                            # same pixels queried
    assert 0.2 < mask.mean() < 0.8
    assert np.max(np.abs(a - fe["run_torso_alpha"])) < 2e-6
    assert np.max(np.abs(c * a + np.float32(1) * (1 - a) - fe["run_torso_bg"])) < 2e-6


def test_torso_grid_update_checker_matches_reference(fe):
    from oracle.occupancy import update_density_grid_torso
    P = FI.torso_weights()
    grid = fe["occ_torso_grid0"].copy()
    mean = update_density_grid_torso(P, grid, fe["torso_enc_anchor"], P["individual_codes_torso"][0], fe["occ_torso_noise"])
    assert np.max(np.abs(grid - fe["occ_torso_grid1"])) < 2e-6
    assert mean == pytest.approx(float(fe["occ_torso_mean"][0]), rel=1e-5)


# ---- head occupancy grid --------------------------------------------------------------------------------------------------------------
def head_params(golden, bound):
    """the generator's head for `bound`: MLP weights of the committed state-dict, tables U(-1, 1) from seed 1234 + bound"""
    spec = TriplaneSpec(float(bound))
    P = {k[3:]: golden[k] for k in golden.files if k.startswith("sd/")}
    rng = np.random.default_rng(1234 + bound)
    for n in ("xy", "yz", "xz"):
        P[f"encoder_{n}.embeddings"] = rng.uniform(-1, 1, (spec.n_params, 1)).astype(np.float32)
        P[f"encoder_{n}.offsets"] = spec.offsets.astype(np.int32)
    return spec, P


@pytest.mark.parametrize("bound", [1, 2])
def test_mark_untrained_grid_checker_matches_reference(fe, bound):
    from oracle.occupancy import mark_untrained_grid
    C = 1 + int(np.ceil(np.log2(bound)))
    grid = FI.initial_density_grid(C)
    count, margin = mark_untrained_grid(grid, fe[f"occ_b{bound}_cams"], fe[f"occ_b{bound}_cam_intr"], bound, return_margin=True)
    ref = fe[f"occ_b{bound}_marked"]
    diff = grid != ref
    assert not (diff & (margin > 1e-5)).any()      # only cells a frustum plane passes through within rounding may differ
    assert diff.mean() < 1e-3 and 0.02 < (ref == -1).mean() < 0.98


@pytest.mark.parametrize("bound", [1, 2])
def test_update_extra_state_checker_matches_reference(fe, golden, bound):
    from oracle.occupancy import update_density_grid
    spec, P = head_params(golden, bound)
    grid = fe[f"occ_b{bound}_marked"].copy()
    eye = np.array([[0.25]], np.float32)
    for it in range(2):
        mean, thresh, bits = update_density_grid(spec, P, grid, fe[f"occ_b{bound}_it{it}_enc_a"], eye, bound, fe[f"occ_b{bound}_it{it}_noise"],
                                                 density_thresh=10)   # Opt.density_thresh (train.py default)
        ref = fe[f"occ_b{bound}_it{it}_grid"]
        assert np.array_equal(grid == -1, ref == -1)                   # untrained cells stay untouched (:763)
        assert np.max(np.abs(grid - ref) / np.maximum(np.abs(ref), 1)) < 2e-5   # the MLP's summation order is free (3e-6 on sigma)
        assert mean == pytest.approx(float(fe[f"occ_b{bound}_it{it}_mean"][0]), rel=1e-5)
        mism = np.unpackbits(bits ^ fe[f"occ_b{bound}_it{it}_bits"], bitorder="little").astype(bool)
        near = (np.abs(ref - mean) < 1e-4 * max(mean, 1e-6)).reshape(-1)
        assert not (mism & ~near).any() and mism.mean() < 1e-3           # threshold = mean density: ties within rounding only
        grid = ref.copy()                                              # continue from the reference's state
