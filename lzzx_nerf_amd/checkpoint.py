"""The reference trainer's checkpoint container, read side (/root/reference/nerf_triplane/TrainerUtil.py:1222-1281 save, :1283-1345 load).

Two file layouts exist in the wild:
  * the container `torch.save` writes at TrainerUtil.py:1253,1278:
        {'epoch', 'global_step', 'stats', 'mean_count', 'mean_density', 'mean_density_torso', 'model': state_dict
         [, 'optimizer', 'lr_scheduler', 'scaler', 'ema' when full=True]}
    where `best` checkpoints drop `model['density_grid']` (:1273-1274) but keep the `density_bitfield` buffer;
  * a bare `state_dict` without a 'model' key, which `load_checkpoint` feeds to `load_state_dict` directly (:1297-1300) -- the three
    running means then keep their constructor values (0: renderer.py:142,149,154).
The model state dict carries, besides the weights, the renderer's buffers (renderer.py:112-153): `aabb_train`, `aabb_infer`,
`density_grid` [cascade, 128^3] (Morton order, -1 = untrained), `density_bitfield` [cascade * 128^3 / 8] uint8, `density_grid_torso`
[128^2] (opt.torso), `step_counter` [16, 2].

What inference needs from it, and where the reference gets it:
  * `density_bitfield`: restored by `load_state_dict`, marched as is (renderer.py:488-530).  When a file lacks it (hand-made state
    dicts) it is what `update_extra_state` would pack: `packbits(density_grid, min(mean_density, density_thresh))`
    (renderer.py:760-766, mean_density = mean of the non-negative cells) -- done with the HIP `packbits` operator; with neither buffer
    every cell is marched (all ones: correct pixels, no empty-space skipping).
  * `mean_count` sizes `march_rays_train`'s sample buffers (renderer.py:287), `mean_density_torso` bounds the torso mask threshold
    `min(density_thresh_torso, mean_density_torso)` (renderer.py:603).
Only this module touches files; everything else takes the parsed state dict.
"""
import math
from typing import Any, Dict, NamedTuple, Optional

import torch


class Checkpoint(NamedTuple):
    model: Dict[str, torch.Tensor]      # the reference's model.state_dict()
    kind: str                           # "container" | "bare"
    mean_count: int
    mean_density: float
    mean_density_torso: float
    epoch: Optional[int]
    global_step: Optional[int]
    extra: Dict[str, Any]               # whatever else the container held (stats, optimizer, ...): untouched


_CONTAINER_KEYS = ("model", "mean_count", "mean_density", "mean_density_torso", "epoch", "global_step")


def _load(f, map_location, weights_only):
    """torch.load with the restricted unpickler; the trainer pickles `stats` into the container (TrainerUtil.py:1227-1231), and with
    use_loss_as_metric off its results are numpy float64 scalars (PSNRMeter.measure: V / N from np.log10), which the restricted unpickler
    rejects -- retried with exactly the numpy scalar / dtype reconstructors allow-listed, still without arbitrary code execution."""
    import pickle
    try:
        return torch.load(f, map_location=map_location, weights_only=weights_only)
    except pickle.UnpicklingError as exc:
        if not weights_only:
            raise
        import numpy as np
        allow = [np.dtype, np.float64, np.float32, np.int64, np.ndarray]
        for mod, names in (("numpy.core.multiarray", ("scalar", "_reconstruct")), ("numpy._core.multiarray", ("scalar", "_reconstruct"))):
            try:
                m = __import__(mod, fromlist=list(names))
                allow += [getattr(m, n) for n in names if hasattr(m, n)]
            except ImportError:
                pass
        allow += [type(np.dtype(t)) for t in ("float64", "float32", "int64")]
        if hasattr(f, "seek"):
            f.seek(0)
        try:
            with torch.serialization.safe_globals(allow):
                return torch.load(f, map_location=map_location, weights_only=True)
        except pickle.UnpicklingError:
            raise RuntimeError("checkpoint holds pickled objects beyond tensors and numpy scalars (%s); for a file you trust pass "
                               "weights_only=False (read_checkpoint / TalkingHeadFrame.from_checkpoint(load_kwargs=dict(weights_only=False)))" % exc) from exc


def read_checkpoint(path_or_dict, map_location="cpu", weights_only=True) -> Checkpoint:
    """Parse either layout (TrainerUtil.py:1295-1312).  `path_or_dict`: a file path / file object for `torch.load`, or the dict itself."""
    obj = path_or_dict
    if not isinstance(obj, dict):
        obj = _load(obj, map_location, weights_only)
    if not isinstance(obj, dict):
        raise RuntimeError("checkpoint does not hold a dict (got %s)" % type(obj).__name__)
    if "model" not in obj:                       # TrainerUtil.py:1297-1300: a bare state dict
        bad = [k for k, v in obj.items() if not torch.is_tensor(v)]
        if bad:
            raise RuntimeError("bare checkpoint holds non-tensor entries %s: neither a state_dict nor the trainer's container" % bad[:4])
        return Checkpoint(dict(obj), "bare", 0, 0.0, 0.0, None, None, {})
    num = lambda v: v.item() if torch.is_tensor(v) else v
    return Checkpoint(dict(obj["model"]), "container", int(num(obj.get("mean_count", 0))), float(num(obj.get("mean_density", 0.0))),
                      float(num(obj.get("mean_density_torso", 0.0))), obj.get("epoch"), obj.get("global_step"),
                      {k: v for k, v in obj.items() if k not in _CONTAINER_KEYS})


def infer_hyper(model: Dict[str, torch.Tensor]) -> Dict[str, Any]:
    """The constructor arguments the tensors themselves determine (the reference re-creates the model from its CLI options and would
    fail in load_state_dict on a mismatch): bound from `aabb_train` (renderer.py:108-112), exp_eye from sigma_net's input width
    (network.py:139), cascade and grid size from `density_grid` / `density_bitfield`."""
    h: Dict[str, Any] = {}
    if "aabb_train" in model:
        h["bound"] = float(model["aabb_train"].reshape(-1)[3])
    w = model.get("sigma_net.net.0.weight")
    if w is not None:
        h["exp_eye"] = int(w.shape[1]) == 36 + 32 + 1
    if "density_grid" in model:
        h["cascade"], cells = int(model["density_grid"].shape[0]), int(model["density_grid"].shape[1])
        h["grid_size"] = round(cells ** (1 / 3))
    elif "density_bitfield" in model and "bound" in h:
        h["cascade"] = 1 + math.ceil(math.log2(max(h["bound"], 1.0)))
        h["grid_size"] = round((int(model["density_bitfield"].numel()) * 8 / h["cascade"]) ** (1 / 3))
    return h


def bitfield_plan(ckpt: Checkpoint, density_thresh: float, source: str = "auto"):
    """Which occupancy bitfield inference marches, as `(source, threshold)`:
        ("bitfield", None)   the stored buffer;
        ("grid", t)          packbits(density_grid, t), t = min(mean_density, density_thresh) (renderer.py:765); the container's
                             mean_density when it has one, else None = "compute mean(clamp(grid, 0)) on device" (renderer.py:760);
        ("ones", None)       neither buffer present.
    `source` forces one of them ("bitfield" / "grid" / "ones") and raises if the file cannot provide it."""
    m = ckpt.model
    has_bits, has_grid = "density_bitfield" in m, "density_grid" in m
    if source == "auto":
        source = "bitfield" if has_bits else ("grid" if has_grid else "ones")
    if source == "bitfield":
        if not has_bits:
            raise RuntimeError("checkpoint has no density_bitfield")
        return "bitfield", None
    if source == "grid":
        if not has_grid:
            raise RuntimeError("checkpoint has no density_grid ('best' checkpoints drop it, TrainerUtil.py:1273-1274)")
        if ckpt.kind == "container" and ckpt.mean_density > 0:
            return "grid", min(ckpt.mean_density, float(density_thresh))
        return "grid", None
    if source == "ones":
        return "ones", None
    raise ValueError("source must be auto / bitfield / grid / ones")


@torch.no_grad()
def resolve_bitfield(ckpt: Checkpoint, density_thresh: float = 10.0, device="cuda", source: str = "auto", grid_size: int = 128):
    """-> (density_bitfield uint8 on `device`, density_grid f32 [cascade, G^3] on `device` or None, plan).  The grid branch runs the HIP
    packbits operator (lz_packbits); there is no host fallback."""
    from . import raymarching as R
    plan = bitfield_plan(ckpt, density_thresh, source)
    m = ckpt.model
    grid = m["density_grid"].to(device=device, dtype=torch.float32).contiguous() if "density_grid" in m else None
    if plan[0] == "bitfield":
        bits = m["density_bitfield"].to(device=device, dtype=torch.uint8).contiguous()
    elif plan[0] == "grid":
        thresh = plan[1]
        if thresh is None:
            thresh = min(float(grid.clamp(min=0).mean()), float(density_thresh))     # renderer.py:760-765
            plan = ("grid", thresh)
        bits = R.packbits(grid, thresh)
    else:
        h = infer_hyper(m)
        cascade = h.get("cascade", 1 + math.ceil(math.log2(max(h.get("bound", 1.0), 1.0))))
        G = h.get("grid_size", grid_size)
        bits = torch.full((cascade * G ** 3 // 8,), 255, dtype=torch.uint8, device=device)
    return bits, grid, plan
