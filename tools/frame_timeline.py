#!/usr/bin/env python3
"""Per-iteration head-launch durations (HIP events on the launch stream) of one bench frame, next to the rows of each launch."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_params, synthetic_camera
from lzzx_nerf_amd.head import FusedTriplaneHead
from lzzx_nerf_amd.renderer import TriplaneRenderer, get_rays

device = torch.device("cuda", 0)
golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
P = make_params(golden)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
bits = dev(np.full(128 ** 3 // 8, 255, np.uint8))
pose, intr = synthetic_camera(512, 512)
rays_o, rays_d = get_rays(dev(pose), intr, 512, 512)
enc_a, ind, eye = dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"])
head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, bound=1.0, device=device)
bf, cap = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 4)
r = TriplaneRenderer(head, bits, bound=1.0, budget_factor=bf, n_step_cap=cap)
f = lambda: r.render(rays_o, rays_d, enc_a, ind, eye, dt_gamma=1 / 256, max_steps=192, T_thresh=1e-4)
for _ in range(3):
    f()
torch.cuda.synchronize()
r.timing_start(256)
o = f()
ms = r.timing_stop()
print("iterations", int(o["state"][6]), "launches timed", len(ms), "sum head ms %.3f" % sum(ms))
print(" ".join("%.0f" % (m * 1e3) for m in ms), "(us per head launch, in order)")
