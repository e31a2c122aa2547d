"""oracle/render.py (the checker's restatement of the reference's render loops) against tests/golden/reference_loops.npz: the outputs of
the reference's OWN run_cuda_for_inference / run_cuda (renderer.py:185-570) and raymarching wrappers (raymarching.py:18-48, 186-280,
347-398, 594-671), run unmodified in the build container on the checker's kernels (tests/golden/make_golden_loops.py).  The same kernels
sit under both sides, so what is compared is everything above them: buffer sizing and padding, the n_step schedule and its cap, mask
compaction, perturb on the first iteration only, abs().sum(-1) ambients, blend / clamp, depth normalisation.

Exact: the (n_alive, n_step, M) of every iteration, per-ray marched counts, the training branch's row counts and counters.  Images and
sums: bit for bit with the reference's own MLP arrangement (head_forward_torch: the same torch ops on the same rows), <= 1e-4 with the
order-pinned head the HIP kernels are held to."""
import os

import numpy as np
import pytest

from conftest import ellipsoid_bitfield
from oracle.head import TriplaneSpec, head_forward_torch
from oracle.render import render_inference, render_train_forward

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INFER = ["ms16", "ms32", "ms64", "ms16_T", "ms32_perturb", "ms24_dg0"]
TRAIN = ["t_all", "t_first", "t_mean4096", "t_mean20000_perturb"]


@pytest.fixture(scope="module")
def loops():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_loops.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def bits():
    return ellipsoid_bitfield()[0]


def _kw(loops, tag):
    ms, dg, T, pert = loops[f"{tag}/kw"]
    return dict(max_steps=int(ms), dt_gamma=float(dg), T_thresh=float(T)), bool(pert)


@pytest.mark.parametrize("tag", INFER)
def test_inference_loop_equals_the_reference_loop(params, loops, bits, tag):
    kw, pert = _kw(loops, tag)
    for path in ("inference", "evaluate"):
        pre = f"{tag}/{path}/"
        noises = loops[pre + "noises"] if pert else None
        st = {}
        got = render_inference(TriplaneSpec(1.0), params, loops["rays_o"], loops["rays_d"], bits, loops["enc_a"], loops["ind_code"], loops["eye"],
                               stats=st, head=head_forward_torch, noises=noises, testing=path == "inference", **kw)
        sched = loops[pre + "schedule"]
        assert [tuple(r) for r in sched[:, :2]] == [tuple(x) for x in st["schedule"]]                       # n_alive, n_step of every iteration
        assert all(int(M) == na * ns + (128 - (na * ns) % 128) for na, ns, M in sched)                      # raymarching.py:379-382
        assert np.array_equal(st["samples_per_ray"], loops[pre + "counts"])
        for k in ("image", "image_raw", "weights_sum", "depth", "amb_aud_sum", "amb_eye_sum", "uncertainty_sum"):
            assert np.array_equal(got[k], loops[pre + k]), (path, k, float(np.abs(got[k] - loops[pre + k]).max()))
        if path == "evaluate":
            d = np.clip(got["depth"] - st["nears"], np.float32(0), None) / (st["fars"] - st["nears"])       # renderer.py:385
            assert np.array_equal(d.astype(np.float32), loops[pre + "depth_norm"], equal_nan=True)
            assert np.array_equal(got["amb_aud_sum"], loops[pre + "ambient_aud"]) and np.array_equal(got["uncertainty_sum"], loops[pre + "uncertainty"])
    # the order-pinned head (what the HIP kernels are bit-equal to): same schedule and counts, image within north_star's bound
    st2 = {}
    pinned = render_inference(TriplaneSpec(1.0), params, loops["rays_o"], loops["rays_d"], bits, loops["enc_a"], loops["ind_code"], loops["eye"],
                              stats=st2, noises=loops[f"{tag}/inference/noises"] if pert else None, **kw)
    assert st2["schedule"] == [tuple(r) for r in loops[f"{tag}/inference/schedule"][:, :2]]
    assert np.array_equal(st2["samples_per_ray"], loops[f"{tag}/inference/counts"])
    assert float(np.abs(pinned["image"] - loops[f"{tag}/inference/image"]).max()) <= 1e-4
    assert float(np.abs(pinned["depth"] - loops[f"{tag}/inference/depth"]).max()) <= 1e-4


def test_the_fixture_holds_the_cap_cases(loops):
    """the reference's deployed cap binds in the fixture: C_eff = sum of n_step = 17 > max_steps = 16, and rays do receive 17 samples"""
    s = loops["ms16/inference/schedule"]
    assert int(s[:, 1].sum()) == 17 and int(loops["ms16/inference/counts"].max()) == 17
    assert int(loops["ms64/inference/counts"].max()) < 64                      # ... and does not at 64
    assert not np.array_equal(loops["ms32_perturb/inference/image"], loops["ms32/inference/image"])
    assert not np.array_equal(loops["ms16_T/inference/counts"], loops["ms16/inference/counts"])     # T_thresh 0.8 cuts rays


@pytest.mark.parametrize("tag", TRAIN)
def test_training_forward_equals_the_reference_run_cuda(params, loops, bits, tag):
    ms, force, mean_count, pert = loops[f"{tag}/kw"]
    got = render_train_forward(TriplaneSpec(1.0), params, loops["rays_o"], loops["rays_d"], bits, loops["enc_a"], loops["ind_code"], loops["eye"],
                               max_steps=int(ms), noises=loops[f"{tag}/noises"] if pert else None, mean_count=int(mean_count),
                               force_all_rays=bool(force), head=head_forward_torch)
    n_rows, M, comp_M = loops[f"{tag}/n_rows"]
    assert got["xyzs"].shape[0] == n_rows == comp_M                                  # the trim / mean_count sizing of raymarching.py:221-256
    assert np.array_equal(got["counter"], loops[f"{tag}/counter"])
    assert np.allclose(got["xyzs"].astype(np.float64).sum(0), loops[f"{tag}/xyzs_sum"], rtol=0, atol=0)
    c = got["comp"]
    for k, ref in (("weights_sum", "weights_sum"), ("amb0_sum", "ambient_aud"), ("amb1_sum", "ambient_eye"), ("unc_sum", "uncertainty")):
        assert np.array_equal(c[k], loops[f"{tag}/{ref}"]), k
    assert np.array_equal(got["image"], loops[f"{tag}/image"])
    assert np.array_equal(got["depth"], loops[f"{tag}/depth_norm"], equal_nan=True)
    if tag == "t_mean4096":
        assert int(got["counter"][0]) > n_rows                                       # the estimate was too small: rays were dropped
        assert float((c["weights_sum"] == 0).mean()) > 0.5
    pinned = render_train_forward(TriplaneSpec(1.0), params, loops["rays_o"], loops["rays_d"], bits, loops["enc_a"], loops["ind_code"], loops["eye"],
                                  max_steps=int(ms), noises=loops[f"{tag}/noises"] if pert else None, mean_count=int(mean_count), force_all_rays=bool(force))
    assert float(np.abs(pinned["image"] - loops[f"{tag}/image"]).max()) <= 1e-4
