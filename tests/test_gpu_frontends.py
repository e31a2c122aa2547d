"""The HIP path of the path's neighbours against the SAME reference-generated vectors the checker is pinned to
(tests/test_golden_frontends.py): ray selection / generation, background coordinates, audio front-end, torso branch,
mark_untrained_grid, update_extra_state (head and torso halves).  Tolerances are the ones stated there (torch fixes no summation
order); against the checker itself these kernels are bit-exact (tests/test_gpu_{parity,audio,torso,occupancy}.py)."""
import os

import numpy as np
import pytest
import torch

import frontends_inputs as FI
from oracle import oracle as O
from test_golden_frontends import RAY_CASES, head_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fe():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_frontends.npz"), allow_pickle=False)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def sd(P):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in P.items()}


@pytest.mark.parametrize("tag,kw", RAY_CASES)
def test_get_rays_reference_signature(fe, tag, kw, monkeypatch):
    from lzzx_nerf_amd import utils as U
    H, W = [int(v) for v in fe["rays_HW"]]
    poses = fe["rays_poses"] if tag != "rect" else fe["rays_poses"][:1]
    ref_inds = fe[f"rays_{tag}_inds"]
    if tag in ("rand", "patch"):   # the CUDA generator draws other pixels than the CPU generator of the fixture: hand it the fixture's draw
        real = U.select_pixels
        drawn = real(H, W, device="cuda", **kw)                                     # still exercise the device-side draw
        assert drawn.shape[0] == ref_inds.shape[1] and int(drawn.min()) >= 0 and int(drawn.max()) < H * W
        if tag == "patch":
            d4 = drawn.view(-1, 4, 4)
            assert bool((d4[:, :, 1:] - d4[:, :, :-1] == 1).all()) and bool((d4[:, 1:, :] - d4[:, :-1, :] == W).all())
        monkeypatch.setattr(U, "select_pixels", lambda *a, **k: dev(ref_inds[0]))
    r = U.get_rays(dev(poses), list(fe["rays_intr"]), H, W, **kw)
    assert set(r) == {"i", "j", "inds", "rays_o", "rays_d"}
    assert tuple(r["inds"].shape) == ref_inds.shape and r["inds"].dtype == torch.int64
    assert np.array_equal(r["inds"].cpu().numpy(), ref_inds)
    for k in ("i", "j", "rays_o"):
        assert np.array_equal(r[k].cpu().numpy(), fe[f"rays_{tag}_{k}"]), k
    assert np.max(np.abs(r["rays_d"].cpu().numpy() - fe[f"rays_{tag}_rays_d"])) < 2e-7
    chk = O.get_rays_batched(poses, fe["rays_intr"], H, W, None if tag == "full" else ref_inds[0])
    assert np.array_equal(r["rays_d"].cpu().numpy(), chk["rays_d"])                  # and bit for bit against the checker
    ro, rd = U.frame_rays(dev(poses[0]), list(fe["rays_intr"]), H, W)
    assert np.array_equal(rd.cpu().numpy(), O.get_rays(poses[0], fe["rays_intr"], H, W)[1])


def test_get_rays_clamps_n_and_rejects_grad(fe):
    from lzzx_nerf_amd.utils import get_rays
    r = get_rays(dev(fe["rays_poses"]), list(fe["rays_intr"]), 48, 64, N=10 ** 6)
    assert tuple(r["rays_d"].shape) == (2, 48 * 64, 3)
    with pytest.raises(RuntimeError, match="gradient"):
        get_rays(dev(fe["rays_poses"]).requires_grad_(True), list(fe["rays_intr"]), 48, 64)


def test_bg_coords(fe):
    from lzzx_nerf_amd.utils import get_bg_coords
    assert np.array_equal(get_bg_coords(48, 64, "cuda").cpu().numpy(), fe["bg_coords_48_64"])
    assert np.array_equal(get_bg_coords(5, 7, "cuda").cpu().numpy(), fe["bg_coords_5_7"])


@pytest.mark.parametrize("dim_in", [29, 44, 1024])
def test_audio_frontend_matches_reference(fe, dim_in):
    from lzzx_nerf_amd.audio import FusedAudioEncoder
    P, a = FI.audio_weights(dim_in), FI.audio_windows(dim_in)
    tol = 3e-6 * max(float(np.abs(fe[f"audio_{dim_in}_feat"]).max()), 1)
    enc = FusedAudioEncoder(sd(P))
    assert np.max(np.abs(enc(dev(a)).cpu().numpy() - fe[f"audio_{dim_in}_enc_a"])) < tol
    noatt = FusedAudioEncoder({k: v for k, v in sd(P).items() if "audio_att_net" not in k})
    assert np.max(np.abs(noatt(dev(a)).cpu().numpy() - fe[f"audio_{dim_in}_feat"])) < tol
    assert np.max(np.abs(noatt(dev(a[:1])).cpu().numpy() - fe[f"audio_{dim_in}_noatt"])) < tol


def test_torso_matches_reference(fe):
    from lzzx_nerf_amd.torso import FusedTorso
    from lzzx_nerf_amd.utils import get_bg_coords
    P = FI.torso_weights()
    torso = FusedTorso(sd(P))
    pose = dev(FI.head_pose()[None])
    ind = dev(P["individual_codes_torso"][0])
    ea = torso.encode_anchor(pose)
    assert np.max(np.abs(ea.cpu().numpy() - fe["torso_enc_anchor"])) < 1e-5
    alpha, color, dx = torso(dev(FI.torso_pixels()), pose, ind)
    assert np.max(np.abs(dx.cpu().numpy() - fe["torso_dx"])) < 5e-6
    assert np.max(np.abs(alpha.cpu().numpy() - fe["torso_alpha"])) < 5e-6 and np.max(np.abs(color.cpu().numpy() - fe["torso_color"])) < 5e-6
    Hh, Ww = [int(v) for v in fe["run_torso_hw"]]
    a, c, _ = torso(get_bg_coords(Hh, Ww, "cuda"), pose, ind, density_grid=dev(fe["run_torso_grid"]), density_thresh=0.01,
                    enc_anchor=dev(fe["torso_enc_anchor"]))
    assert np.array_equal(a.cpu().numpy()[:, 0] != 0, fe["run_torso_alpha"][:, 0] != 0)      # same pixels pass the 2-D occupancy mask
    assert np.max(np.abs(a.cpu().numpy() - fe["run_torso_alpha"])) < 5e-6
    assert np.max(np.abs(FusedTorso.mix_background(a, c, 1.0).cpu().numpy() - fe["run_torso_bg"])) < 5e-6


def test_torso_grid_update_matches_reference(fe):
    from lzzx_nerf_amd.occupancy import update_density_grid_torso
    from lzzx_nerf_amd.torso import FusedTorso
    P = FI.torso_weights()
    torso = FusedTorso(sd(P))
    grid = dev(fe["occ_torso_grid0"].copy())
    mean, _ = update_density_grid_torso(torso, grid, None, dev(P["individual_codes_torso"][0]), noise=dev(fe["occ_torso_noise"]),
                                        enc_anchor=dev(fe["torso_enc_anchor"]))
    assert np.max(np.abs(grid.cpu().numpy() - fe["occ_torso_grid1"])) < 5e-6
    assert float(mean) == pytest.approx(float(fe["occ_torso_mean"][0]), rel=1e-5)


@pytest.mark.parametrize("bound", [1, 2])
def test_mark_untrained_grid(fe, bound):
    from lzzx_nerf_amd.occupancy import mark_untrained_grid
    from oracle.occupancy import mark_untrained_grid as oracle_mark
    C = 1 + int(np.ceil(np.log2(bound)))
    g0 = FI.initial_density_grid(C)
    cams, intr = fe[f"occ_b{bound}_cams"], fe[f"occ_b{bound}_cam_intr"]
    ref = g0.copy()
    count_o, margin = oracle_mark(ref, cams, intr, bound, return_margin=True)
    grid = dev(g0.copy())
    count = mark_untrained_grid(grid, dev(cams), list(intr), bound=float(bound), return_count=True)
    assert np.array_equal(count.cpu().numpy(), count_o)                  # camera counts per cell: bit for bit against the checker
    assert np.array_equal(grid.cpu().numpy(), ref)
    diff = grid.cpu().numpy() != fe[f"occ_b{bound}_marked"]              # and against the reference's Python up to rounding ties
    assert not (diff & (margin > 1e-5)).any()


@pytest.mark.parametrize("bound", [1, 2])
def test_update_extra_state_matches_reference(fe, golden, bound):
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.occupancy import update_density_grid
    _, P = head_params(golden, bound)
    head = FusedTriplaneHead(sd(P), bound=float(bound))
    C = 1 + int(np.ceil(np.log2(bound)))
    grid = dev(fe[f"occ_b{bound}_marked"].copy())
    bits = torch.zeros(C * 16 ** 3 // 8, dtype=torch.uint8, device="cuda")
    eye = dev(np.array([[0.25]], np.float32))
    for it in range(2):
        mean, thresh = update_density_grid(head, grid, bits, dev(fe[f"occ_b{bound}_it{it}_enc_a"]), eye, bound=float(bound),
                                           density_thresh=10, noise=dev(fe[f"occ_b{bound}_it{it}_noise"]))
        ref = fe[f"occ_b{bound}_it{it}_grid"]
        g = grid.cpu().numpy()
        assert np.array_equal(g == -1, ref == -1)
        assert np.max(np.abs(g - ref) / np.maximum(np.abs(ref), 1)) < 2e-5
        m = float(mean)
        assert m == pytest.approx(float(fe[f"occ_b{bound}_it{it}_mean"][0]), rel=1e-5) and float(thresh) == m
        mism = np.unpackbits(bits.cpu().numpy() ^ fe[f"occ_b{bound}_it{it}_bits"], bitorder="little").astype(bool)
        near = (np.abs(ref - m) < 1e-4 * max(m, 1e-6)).reshape(-1)
        assert not (mism & ~near).any()
        grid.copy_(dev(ref))
