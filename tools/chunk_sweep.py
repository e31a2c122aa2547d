#!/usr/bin/env python3
"""Host-side loop parameters of TriplaneRenderer (iterations per C call, chunks of look-ahead) on the bench frame."""
import os, sys, time
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
for chunk, look in ((8, 2), (4, 2), (4, 1), (2, 2), (2, 3), (3, 2), (6, 1), (16, 1), (32, 0)):
    r = TriplaneRenderer(head, bits, bound=1.0, budget_factor=4, n_step_cap=4)
    r.chunk, r.lookahead = chunk, look
    f = lambda: r.render(rays_o, rays_d, enc_a, ind, eye, dt_gamma=1 / 256, max_steps=192, T_thresh=1e-4)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        o = f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"chunk {chunk:2d} lookahead {look}: {dt * 1e3:7.3f} ms/frame  {int(o['state'][5]) / dt / 1e9:.3f} Gsamples/s", flush=True)
