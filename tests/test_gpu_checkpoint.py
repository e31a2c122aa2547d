"""A frame rendered from a checkpoint FILE in the reference trainer's layout (TrainerUtil.py:1222-1345) equals the frame rendered from the
same tensors handed over directly, for every layout the trainer writes: full container, `best` (no density_grid), bare state dict, and
files without the bitfield buffer (rebuilt with the HIP packbits at min(mean_density, density_thresh), renderer.py:760-766)."""
import numpy as np
import pytest
import torch

from conftest import synthetic_camera

pytestmark = pytest.mark.gpu


def _setup(params, golden):
    from lzzx_nerf_amd.synthetic import ellipsoid_bitfield_device
    from lzzx_nerf_amd.utils import frame_rays
    from test_audio_oracle import audio_state
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}
    sd.update({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in audio_state(29, 32, True).items()})
    bits, grid = ellipsoid_bitfield_device("cuda")                 # grid: 1 inside the ellipsoid, 0 outside
    grid = grid * 40.0                                             # densities: mean 40 * fill fraction
    grid[0, ::11] = -1.0                                           # untrained cells stay unoccupied for any threshold >= 0
    sd["density_grid"] = grid.cpu()
    sd["aabb_train"] = torch.tensor([-1.0, -0.5, -1.0, 1.0, 0.5, 1.0])         # renderer.py:110: y extent halved
    sd["aabb_infer"] = sd["aabb_train"].clone()
    sd["step_counter"] = torch.zeros(16, 2, dtype=torch.int32)
    H = W = 40
    pose, intr = synthetic_camera(H, W)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ro, rd = frame_rays(dev(pose), intr, H, W)
    auds = dev(np.random.default_rng(3).normal(size=(8, 29, 16)).astype(np.float32))
    args = (ro, rd, auds)
    kw = dict(eye=dev(golden["net_eye"]), ind_code=dev(golden["net_ind"]), max_steps=48, count_samples=True)
    return sd, grid, args, kw


def _same(a, b):
    return all(torch.equal(a[k], b[k]) for k in ("image", "depth", "ray_counts"))


def test_frame_from_checkpoint_files_equals_direct_path(params, golden, tmp_path):
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd.pipeline import TalkingHeadFrame
    sd, grid, args, kw = _setup(params, golden)
    mean_density = float(grid.clamp(min=0).mean())
    thresh = min(mean_density, 10.0)
    bits = R.packbits(grid, thresh)                                 # what update_extra_state left in the buffer (renderer.py:765-766)
    assert 0 < int(bits.count_nonzero()) < bits.numel()
    sd["density_bitfield"] = bits.cpu()
    direct = TalkingHeadFrame(sd, bits, bound=1.0).render(*args, **kw)
    assert int(direct["ray_counts"].sum()) > 1000
    container = {"epoch": 3, "global_step": 99, "stats": {"loss": [0.5], "checkpoints": []}, "mean_count": 2048, "mean_density": mean_density,
                 "mean_density_torso": 0.0, "model": sd}
    best = dict(container, model={k: v for k, v in sd.items() if k != "density_grid"})
    no_bits = dict(container, model={k: v for k, v in sd.items() if k != "density_bitfield"})
    files = {"full": container, "best": best, "bare": sd, "no_bits": no_bits, "bare_no_bits": no_bits["model"]}
    for name, obj in files.items():
        path = tmp_path / (name + ".pth")
        torch.save(obj, path)
        f = TalkingHeadFrame.from_checkpoint(str(path), density_thresh=10.0)
        assert f.checkpoint_kind == ("bare" if name.startswith("bare") else "container"), name
        assert f.bitfield_plan[0] == ("grid" if "no_bits" in name else "bitfield"), name
        if "no_bits" in name:                                        # rebuilt: same threshold rule, same HIP packbits -> same bytes
            assert f.bitfield_plan[1] == pytest.approx(thresh, rel=1e-6) and torch.equal(f.renderer.bitfield, bits), name
        assert (f.density_grid is None) == (name == "best")
        if name in ("full", "best", "no_bits"):
            assert (f.mean_count, f.epoch, f.global_step) == (2048, 3, 99) and f.mean_density == pytest.approx(mean_density)
        else:
            assert (f.mean_count, f.mean_density) == (0, 0.0)
        assert _same(f.render(*args, **kw), direct), name
    # forced rebuild from the grid of a full checkpoint, and a lower density_thresh marks more cells
    f = TalkingHeadFrame.from_checkpoint(container, density_thresh=10.0, bitfield="grid")
    assert torch.equal(f.renderer.bitfield, bits)
    f = TalkingHeadFrame.from_checkpoint(container, density_thresh=1e-3, bitfield="grid")
    assert int(f.renderer.bitfield.count_nonzero()) >= int(bits.count_nonzero())
    # neither buffer: every cell is marched; pixels of rays that hit the ellipsoid still come from the same samples plus empty ones
    bare_min = {k: v for k, v in sd.items() if k not in ("density_grid", "density_bitfield")}
    f = TalkingHeadFrame.from_checkpoint(bare_min)
    assert f.bitfield_plan == ("ones", None) and bool((f.renderer.bitfield == 255).all())
    out = f.render(*args, **kw)
    assert int(out["ray_counts"].sum()) > int(direct["ray_counts"].sum())
    with pytest.raises(RuntimeError, match="best"):
        TalkingHeadFrame.from_checkpoint(best, bitfield="grid")


def test_checkpoint_hyper_parameters_reach_the_head(params, golden):
    """exp_eye off (sigma_net sees 68 inputs) and bound come from the tensors; an explicit argument still wins"""
    from lzzx_nerf_amd.pipeline import TalkingHeadFrame
    sd, grid, args, kw = _setup(params, golden)
    sd["sigma_net.net.0.weight"] = sd["sigma_net.net.0.weight"][:, :68].contiguous()
    kw = dict(kw, eye=None)
    f = TalkingHeadFrame.from_checkpoint(sd)
    assert f.head.has_eye is False and f.renderer.bound == 1.0
    g = TalkingHeadFrame(sd, f.renderer.bitfield, bound=1.0, exp_eye=False)
    assert _same(f.render(*args, **kw), g.render(*args, **kw))
