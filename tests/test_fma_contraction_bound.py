"""How much can the checker's reading of nvcc's FMA contraction matter?  

oracle/*.c assume that nvcc's default -fmad=true fuses `a * b + c` written as ONE expression and nothing else; NVPTX-style fusion may
also fuse across statements (`weight = alpha * T; weight_sum += weight;`, raymarching.cu:2203-2206).  Without nvcc that cannot be decided,
so a second build of the same checker sources takes the other side everywhere it can (gcc -ffp-contract=fast -mfma, plus the compositing
accumulation fused explicitly: oracle/Makefile `fast`), and this test renders whole frames with both and reports the distance:
pixels above 1e-6, the largest difference, rays whose sample count changes.  The bound that must hold is north_star's 1e-4 on the image."""
import json
import os

import numpy as np
import pytest

from conftest import ellipsoid_bitfield, synthetic_camera
from oracle import oracle as O
from oracle.head import TriplaneSpec, get_rays
from oracle.render import render_inference


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


@pytest.mark.skipif(not _has_fma(), reason="the contracted build needs a CPU with FMA")
@pytest.mark.parametrize("scene,max_steps,T_thresh", [("ellipsoid", 64, 1e-4), ("ones", 48, 1e-4), ("ellipsoid", 16, 1e-4),
                                                      ("ones", 96, 0.6), ("ellipsoid", 64, 0.7)])      # the last two: T_thresh cuts most rays
def test_the_contraction_model_moves_the_image_by_far_less_than_the_parity_bound(params, golden, scene, max_steps, T_thresh):
    fast = O.build_fast()
    H = W = 64
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = np.full(128 ** 3 // 8, 255, np.uint8) if scene == "ones" else ellipsoid_bitfield()[0]
    args = (TriplaneSpec(1.0), params, ro, rd, bits, golden["net_enc_a"], golden["net_ind"], golden["net_eye"])
    sa, sb = {}, {}
    a = render_inference(*args, stats=sa, max_steps=max_steps, T_thresh=T_thresh)
    with O.variant(fast):
        b = render_inference(*args, stats=sb, max_steps=max_steps, T_thresh=T_thresh)
    d = np.abs(a["image"].astype(np.float64) - b["image"])
    dd = np.abs(a["depth"].astype(np.float64) - b["depth"])
    flips = int((sa["samples_per_ray"] != sb["samples_per_ray"]).sum())
    report = dict(scene=scene, max_steps=max_steps, T_thresh=T_thresh, pixels=H * W, rays_cut_by_T=int((a['weights_sum'] > 1 - T_thresh).sum()), pixels_above_1e6=int((d.max(1) > 1e-6).sum()), max_abs_image=float(d.max()),
                  max_abs_depth=float(dd.max()), rays_with_another_sample_count=flips, schedule_equal=sa["schedule"] == sb["schedule"],
                  bit_identical_pixels=int((d.max(1) == 0).sum()))
    print("fma-contraction bound:", json.dumps(report))
    assert d.max() <= 1e-4 and dd.max() <= 1e-4                      # north_star's image bound holds under either reading
    assert d.max() > 0 or scene == "never"                           # the two builds ARE different arithmetic (else the experiment says nothing)
    out = os.environ.get("LZ_FMA_REPORT")
    if out:
        with open(out, "a") as f:
            f.write(json.dumps(report) + "\n")
