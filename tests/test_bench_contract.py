"""bench.py's stdout contract line stays small enough for the driver to parse (round 4's grew to 21.9 KB: BENCH_r04.parsed = null) and
keeps exactly the keys the contract names; everything else lives in bench_detail.json.  CPU only: tools/bench_contract.compact is pure."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_contract import (CONFIG_KEYS, CONTRACT_KEYS, CPU_KEYS, MAX_LINE_BYTES, PARITY_KEYS, ROOFLINE_KEYS,  # noqa: E402
                                  TARGET_LINE_BYTES, compact)


def _canned():
    """round 4's full 21.9 KB result: the largest line bench.py has ever produced"""
    return json.load(open(os.path.join(ROOT, "profiles", "r4_final_bench.json")))


def test_contract_line_is_small_and_complete():
    full = _canned()
    assert len(json.dumps(full)) > 20000            # the canned input really is the oversize one
    line = compact(full)
    s = json.dumps(line)
    assert len(s) < TARGET_LINE_BYTES < MAX_LINE_BYTES, len(s)
    assert "\n" not in s
    for k in CONTRACT_KEYS:
        assert k in line and (line[k] is not None or k == "vs_baseline"), k
    assert set(line["config"]) == set(CONFIG_KEYS) and "model" not in line["config"]
    for r in ("roofline", "roofline_f16"):
        assert set(ROOFLINE_KEYS) <= set(line[r]), r
        assert line[r]["bound"] == "mfma" and 0 < line[r]["frac"] < 1 and line[r]["avg_launch_ms"] > 0
    assert line["roofline_gridencoder"]["bound"] == "hbm" and line["roofline_gridencoder"]["unit"] == "GB/s"
    assert set(CPU_KEYS) <= set(line["cpu_baseline"]) and line["cpu_baseline"]["kind"] in ("port", "reference")
    assert line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["s_per_frame"] > 0
    for k in PARITY_KEYS:
        assert k in line
    assert line["dtype"] == "f32" and line["metric"].startswith("rendered samples/s")
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["detail"] == "bench_detail.json"


def test_contract_line_survives_a_pathological_result():
    """side legs of any size never reach the line; error strings are clipped; a result with no side legs still yields the contract"""
    full = _canned()
    full["leg_errors"] = {f"leg{i}": "x" * 5000 for i in range(40)}
    full["some_new_leg"] = {"blob": "y" * 100000}
    full["config"]["workload"] = "w" * 10000
    full["n_gpus"] = 8
    full["tiles_contiguous"] = dict(value=1.0, unit="samples/s", ms_per_step=1.0, scaling="strong", frames_per_step=1, parallelism="p" * 9999)
    s = json.dumps(compact(full))
    assert len(s) < MAX_LINE_BYTES, len(s)
    bare = {k: full[k] for k in CONTRACT_KEYS}
    bare["roofline"] = full["roofline"]
    line = compact(bare)
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and "cpu_baseline" not in line and len(json.dumps(line)) < 2048


def test_bench_prints_through_the_contract_module():
    """bench.py's only stdout JSON of a frame run is compact(result); the side legs are written to the detail file"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "json.dumps(compact(result, detail))" in src and "print(json.dumps(result))" not in src
    assert src.count("print(") <= 3, "bench.py prints: the train-only object, the contract line -- nothing else on stdout"
