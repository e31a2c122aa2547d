"""The reference trainer's checkpoint container (TrainerUtil.py:1222-1345), read side, without a GPU: both file layouts parse, the running
means travel, and the choice of the occupancy bitfield follows the reference (stored buffer; else packbits(density_grid, min(mean_density,
density_thresh)), renderer.py:760-766; else all ones).  The device half (HIP packbits, a frame rendered from a file) is tests/test_gpu_checkpoint.py."""
import numpy as np
import pytest
import torch

from lzzx_nerf_amd.checkpoint import bitfield_plan, infer_hyper, read_checkpoint


def reference_state_dict(params, cascade=1, G=16, torso=False, seed=0):
    """a state dict with the reference's keys (tests/golden: sd_keys) -- weights from the fixture, renderer buffers synthetic"""
    rng = np.random.default_rng(seed)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}
    grid = rng.uniform(-0.5, 30.0, (cascade, G ** 3)).astype(np.float32)
    grid[:, ::7] = -1.0                                                        # untrained cells (renderer.py:691)
    sd["density_grid"] = torch.from_numpy(grid)
    sd["density_bitfield"] = torch.from_numpy(rng.integers(0, 256, cascade * G ** 3 // 8, dtype=np.uint8))
    sd["aabb_train"] = torch.tensor([-1.0, -0.5, -1.0, 1.0, 0.5, 1.0])
    sd["aabb_infer"] = sd["aabb_train"].clone()
    sd["step_counter"] = torch.zeros(16, 2, dtype=torch.int32)
    if torso:
        sd["density_grid_torso"] = torch.from_numpy(rng.uniform(0, 0.1, 128 * 128).astype(np.float32))
    return sd


def container(sd, best=False, full=False, **means):
    """what save_checkpoint writes (TrainerUtil.py:1229-1278)"""
    state = {"epoch": 7, "global_step": 1234, "stats": {"loss": [0.1, 0.05], "valid_loss": [], "results": [], "checkpoints": ["a.pth"], "best_result": None},
             "mean_count": means.get("mean_count", 4321), "mean_density": means.get("mean_density", 3.25),
             "mean_density_torso": means.get("mean_density_torso", 0.004)}
    if full:
        state["optimizer"] = {"state": {}, "param_groups": [{"lr": 1e-3}]}
        state["scaler"] = {"scale": 65536.0}
    model = dict(sd)
    if best:
        del model["density_grid"]                                              # TrainerUtil.py:1273-1274
    state["model"] = model
    return state


def test_container_and_bare_layouts_roundtrip_through_files(params, tmp_path):
    sd = reference_state_dict(params)
    full = tmp_path / "ngp_ep0007.pth"
    bare = tmp_path / "bare.pth"
    best = tmp_path / "ngp.pth"
    torch.save(container(sd, full=True), full)
    torch.save(sd, bare)
    torch.save(container(sd, best=True), best)
    c = read_checkpoint(str(full))
    assert c.kind == "container" and (c.mean_count, c.mean_density, c.mean_density_torso) == (4321, 3.25, 0.004)
    assert (c.epoch, c.global_step) == (7, 1234) and "optimizer" in c.extra and "stats" in c.extra and "model" not in c.extra
    assert set(c.model) == set(sd) and all(torch.equal(c.model[k], sd[k]) for k in sd)
    b = read_checkpoint(str(bare))
    assert b.kind == "bare" and (b.mean_count, b.mean_density, b.mean_density_torso) == (0, 0.0, 0.0) and b.epoch is None
    assert set(b.model) == set(sd)
    s = read_checkpoint(str(best))
    assert "density_grid" not in s.model and "density_bitfield" in s.model
    # a dict handed over directly is the same thing
    assert read_checkpoint(container(sd)).mean_count == 4321
    # tensors as running means (a trainer that kept them on device)
    assert read_checkpoint(container(sd, mean_density=torch.tensor(2.5), mean_count=torch.tensor(9))).mean_density == 2.5


def test_rejects_what_is_neither_layout(tmp_path):
    p = tmp_path / "junk.pth"
    torch.save([1, 2, 3], p)
    with pytest.raises(RuntimeError, match="does not hold a dict"):
        read_checkpoint(str(p))
    with pytest.raises(RuntimeError, match="neither a state_dict"):
        read_checkpoint({"epoch": 3, "weights": torch.zeros(2)})


def test_bitfield_plan_follows_the_reference(params):
    sd = reference_state_dict(params)
    assert bitfield_plan(read_checkpoint(container(sd)), 10.0) == ("bitfield", None)            # load_state_dict restores the buffer
    assert bitfield_plan(read_checkpoint(container(sd, best=True)), 10.0) == ("bitfield", None)
    no_bits = {k: v for k, v in sd.items() if k != "density_bitfield"}
    assert bitfield_plan(read_checkpoint(container(no_bits)), 10.0) == ("grid", 3.25)            # min(mean_density, density_thresh)
    assert bitfield_plan(read_checkpoint(container(no_bits)), 1.5) == ("grid", 1.5)
    assert bitfield_plan(read_checkpoint(no_bits), 10.0) == ("grid", None)                       # bare: mean computed from the grid
    assert bitfield_plan(read_checkpoint(container(sd)), 10.0, "grid") == ("grid", 3.25)         # forced rebuild
    neither = {k: v for k, v in no_bits.items() if k != "density_grid"}
    assert bitfield_plan(read_checkpoint(neither), 10.0) == ("ones", None)
    with pytest.raises(RuntimeError, match="best"):
        bitfield_plan(read_checkpoint(container(sd, best=True)), 10.0, "grid")
    with pytest.raises(RuntimeError, match="no density_bitfield"):
        bitfield_plan(read_checkpoint(neither), 10.0, "bitfield")


def test_hyper_parameters_from_tensors(params):
    sd = reference_state_dict(params, cascade=1, G=16)
    h = infer_hyper(sd)
    assert h == {"bound": 1.0, "exp_eye": True, "cascade": 1, "grid_size": 16}
    sd2 = dict(sd)
    sd2["sigma_net.net.0.weight"] = sd["sigma_net.net.0.weight"][:, :68].contiguous()
    sd2["aabb_train"] = torch.tensor([-2.0, -1.0, -2.0, 2.0, 1.0, 2.0])
    del sd2["density_grid"]
    sd2["density_bitfield"] = torch.zeros(2 * 128 ** 3 // 8, dtype=torch.uint8)
    assert infer_hyper(sd2) == {"bound": 2.0, "exp_eye": False, "cascade": 2, "grid_size": 128}


def test_container_with_numpy_scalars_in_stats_loads_under_the_restricted_unpickler(params, tmp_path):
    """use_loss_as_metric off: stats['results'] holds np.float64 values (PSNRMeter.measure = V / N built from np.log10), which
    torch.load(weights_only=True) rejects; read_checkpoint retries with the numpy scalar reconstructors allow-listed"""
    sd = reference_state_dict(params)
    state = container(sd)
    state["stats"]["results"] = [np.float64(31.25), np.float64(32.5)]
    state["stats"]["best_result"] = np.float64(32.5)
    path = tmp_path / "np_stats.pth"
    torch.save(state, path)
    with pytest.raises(Exception):
        torch.load(str(path), weights_only=True)                     # the premise: the plain restricted load fails on this file
    c = read_checkpoint(str(path))
    assert c.kind == "container" and c.mean_count == 4321 and float(c.extra["stats"]["best_result"]) == 32.5
    assert read_checkpoint(str(path), weights_only=False).mean_count == 4321
    with open(path, "rb") as fh:                                      # file objects are rewound for the retry
        assert read_checkpoint(fh).epoch == 7
