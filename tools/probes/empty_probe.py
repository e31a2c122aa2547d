"""every renderer on a batch of NO rays (a rank whose tile of a small frame is empty): empty outputs, no error -- run on the GPU box"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from lzzx_nerf_amd.synthetic import load_golden, make_params, ellipsoid_bitfield_device
from lzzx_nerf_amd.head import FusedTriplaneHead
from lzzx_nerf_amd.renderer import TriplaneRenderer
g = load_golden(); P = make_params(g)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bits, _ = ellipsoid_bitfield_device("cuda")
head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, bound=1.0)
cond = (dev(g["net_enc_a"]), dev(g["net_ind"]), dev(g["net_eye"]))
e = torch.empty(0, 3, device="cuda")
bad = 0
for mode in ("loop", "fused"):
    for cap in ("reference", "per_ray"):
        try:
            r = TriplaneRenderer(head, bits, bound=1.0, mode=mode, cap=cap)
            o = r.render(e, e, *cond, max_steps=16, count_samples=True)
            torch.cuda.synchronize()
            print(mode, cap, "ok", {k: tuple(v.shape) for k, v in o.items() if k in ("image", "ray_counts")})
        except Exception as ex:
            bad += 1
            print(mode, cap, "FAIL", type(ex).__name__, str(ex)[:200])
sys.exit(bad)
