"""The ONE stdout line of bench.py: a compact object (target <= 4 KB, hard bound MAX_LINE_BYTES) holding exactly the driver's contract --
metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config, the two
roofline objects (f32 headline kernel, f16 = the reference's deployed arithmetic), the grid-encoder roofline north_star asks for, the CPU
baseline and the parity scalars.  Everything else bench.py measures (side legs, schedule prose, per-leg workloads) lives in
bench_detail.json, written next to bench.py; `detail` names it.

Round 4's line had grown to 21.9 KB and the driver could no longer parse it (Weak 1): tests/test_bench_contract.py bounds the
size on a canned full-size result and tests/test_gpu_cfg4.py on the real line."""
import json

MAX_LINE_BYTES = 8192
TARGET_LINE_BYTES = 4096

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")
CONFIG_KEYS = ("workload", "rays_per_step", "rays_per_rank", "samples_per_step", "parallelism", "mode")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel", "avg_launch_ms")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "s_per_frame", "runs")
PARITY_KEYS = ("psnr_vs_checker_db", "max_abs_diff_vs_checker", "sample_counts_equal", "parity_sample")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _roofline(r):
    if not r:
        return None
    out = {k: r.get(k) for k in ROOFLINE_KEYS}
    out["kernel"] = _short(out["kernel"], 48) if out["kernel"] else None
    return out


def _pick(d, *path):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return None
        d = d[p]
    return d


def compact(result, detail="bench_detail.json"):
    """full result dict of bench.main() -> the contract object.  Pure: no GPU, no files (tests call it on canned results)."""
    line = {k: result.get(k) for k in CONTRACT_KEYS}
    cfg = result.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in CONFIG_KEYS}
    line["config"]["workload"] = _short(line["config"]["workload"], 200)
    line["config"]["parallelism"] = _short(line["config"]["parallelism"], 120)
    line["dtype"] = _short(line["dtype"], 16).split(" ")[0] if line["dtype"] else None
    line["roofline"] = _roofline(result.get("roofline"))
    f16 = result.get("roofline_f16") or _pick(result, "fp16_head", "roofline")
    if f16:
        line["roofline_f16"] = _roofline(f16)
        ms16 = _pick(result, "fp16_head", "ms_per_step")
        if ms16 is not None:
            line["roofline_f16"]["ms_per_step"] = ms16
            line["roofline_f16"]["value"] = _pick(result, "fp16_head", "value")
    g = _pick(result, "roofline_gridencoder", "triplane_plane_D2_L12_C1_f32")
    if g:
        line["roofline_gridencoder"] = {k: g.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        line["roofline_gridencoder"]["kernel"] = "lz_k_grid_forward_lds<float,2,1> (triplane plane, 248 B/sample)"
    cpu = result.get("cpu_baseline")
    if cpu:
        c = {k: cpu.get(k) for k in CPU_KEYS}
        leg = next((v for k, v in (cpu.get("legs") or {}).items() if k.startswith("cfg3")), {})
        c["s_per_frame"] = c["s_per_frame"] if c["s_per_frame"] is not None else leg.get("s_per_frame")
        c["runs"] = c["runs"] if c["runs"] is not None else leg.get("runs")
        c["sample"] = _short(c["sample"], 160)
        line["cpu_baseline"] = c
    for k in PARITY_KEYS:
        if k in result:
            line[k] = result[k]
    # the other BASELINE configs, one scalar each (their objects are in the detail file)
    legs = {}
    for name, path in (("train_step_ms", ("train_step", "ms_per_step_median")), ("train_step_f16_ms", ("train_step_f16", "ms_per_step_median")),
                       ("train_full_frame_ms", ("train_step_full_frame", "ms_per_step_median")),
                       ("train_full_frame_f16_ms", ("train_step_full_frame_f16", "ms_per_step_median")),
                       ("cfg2_fused_ms", ("cfg2_hashgrid_render", "f32_tables", "fused", "ms_per_frame")),
                       ("cfg2_fused_f16_ms", ("cfg2_hashgrid_render", "f16_tables", "fused", "ms_per_frame")),
                       ("cfg5_f16_ms", ("cfg5_1024_ellipsoid_f16", "ms_per_step")),
                       ("reference_schedule_loop_ms", ("reference_schedule", "ms_per_step")),
                       ("deployed_max_steps_16_ms", ("deployed_max_steps_16", "ms_per_step")),
                       ("deployed_max_steps_16_f16_ms", ("deployed_max_steps_16", "f16_ms_per_step")),
                       ("talking_head_frame_f16_ms", ("talking_head_frame", "f16", "ms_per_frame"))):
        v = _pick(result, *path)
        if v is not None:
            legs[name] = v
    if legs:
        line["legs_ms"] = legs
    checks = {}
    for name, path in (("reference_schedule_image_equal", ("reference_schedule", "image_equal_to_headline_schedule")),
                       ("deployed_16_image_equal", ("deployed_max_steps_16", "image_equal_to_reference_schedule")),
                       ("deployed_16_counts_equal", ("deployed_max_steps_16", "ray_counts_equal_to_reference_schedule"))):
        v = _pick(result, *path)
        if v is not None:
            checks[name] = v
    if checks:
        line["checks"] = checks
    # N > 1: what lets the first SCALE record validate itself
    for k in ("gathered_frame_ok", "gathered_frame_equals_unsharded_reference_loop", "verify_frames"):
        if k in result:
            line[k] = result[k]
    if result.get("n_gpus", 1) > 1:
        for k in ("tiles_contiguous", "tiles_interleaved", "clip_weak_scaling"):
            if k in result:
                line[k] = {kk: result[k].get(kk) for kk in ("value", "unit", "ms_per_step", "scaling", "frames_per_step")}
        if "rank_tile_ms" in result:
            line["rank_tile_ms"] = result["rank_tile_ms"]
    if result.get("leg_errors"):
        line["leg_errors"] = {k: _short(v, 80) for k, v in list(result["leg_errors"].items())[:6]}
    line["detail"] = detail
    n = len(json.dumps(line))
    if n > MAX_LINE_BYTES:      # never print an unparseable line: drop the optional blocks, the contract itself is < 2 KB
        for k in ("legs_ms", "checks", "leg_errors", "roofline_gridencoder", "tiles_contiguous", "tiles_interleaved", "clip_weak_scaling"):
            line.pop(k, None)
    return line
