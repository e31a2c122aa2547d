#!/usr/bin/env python3
"""BASELINE cfg2 alone (256 x 256 rays, max_steps 128, all-ones occupancy, hash-grid NeRF): the fused path (lzzx_nerf_amd/ngp.py) under a
few schedules, next to the operator-API device loop.  tools/cfg2_bench.py [f32|f16] [--ref] [--sched B,C ...]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lzzx_nerf_amd.ngp import FusedHashgridNeRF, HashgridRenderer
from lzzx_nerf_amd.synthetic import GenericHashgridNeRF, synthetic_camera
from lzzx_nerf_amd.utils import frame_rays

half = len(sys.argv) > 1 and sys.argv[1] == "f16"
dev = torch.device("cuda", 0)
pose, intr = synthetic_camera(256, 256)
ro, rd = frame_rays(torch.from_numpy(np.ascontiguousarray(pose)).to(dev), intr, 256, 256)
aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device=dev)
bits = torch.full((128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
g = GenericHashgridNeRF(dev, half_tables=half)
net = FusedHashgridNeRF(g.enc, g.sigma_net, g.color_net, half_tables=half)


def timed(f, n=10):
    for _ in range(3):
        o = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        o = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, o


scheds = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:] if "," in a] or [(8, 8), (4, 4), (8, 16), (16, 16), (1, 8)]
out = {}
for s in scheds:
    r = HashgridRenderer(net, bits, bound=1.0, aabb=aabb, budget_factor=s[0], n_step_cap=s[1])
    ms, o = timed(lambda: r.render(ro, rd, max_steps=128))
    st = o["state"].cpu().numpy()
    out["%dx%d" % s] = dict(ms=round(ms, 3), samples=int(st[5]), rows=int(st[72]), iterations=int(st[6]))
    del r
if "--ref" in sys.argv:
    from lzzx_nerf_amd.renderer import NetworkRenderer
    nr = NetworkRenderer(lambda x, d: g.net(x, d, 1.0), bits, bound=1.0, aabb=aabb, graph=True)
    ms, o = timed(lambda: nr.render(ro, rd, max_steps=128), 5)
    out["operator_api_device_loop_hipgraph"] = dict(ms=round(ms, 3))
print(json.dumps(out))
