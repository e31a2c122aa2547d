"""the table scatter of the cfg3 training step alone: lz_grid_encode_backward per plane on the step's samples (65 536 random rays, ones
grid), in either sample-row layout -- tools/scatter_bench.py [ray|step]; LZZX_NERF_HIP_SO selects a library variant"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_params
from lzzx_nerf_amd import raymarching as R
from lzzx_nerf_amd._util import call, ptr, stream
from lzzx_nerf_amd.head_train import FusedTriplaneTrainHead
from lzzx_nerf_amd.synthetic import ones_bitfield, synthetic_camera
from lzzx_nerf_amd.utils import frame_rays

layout = sys.argv[1] if len(sys.argv) > 1 else "step"
dev = torch.device("cuda")
golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
net = FusedTriplaneTrainHead(make_params(golden), bound=1.0, forward_dtype="f16", backward_dtype="f16").to(dev)
g = torch.Generator(device=dev).manual_seed(0)
pose, intr = synthetic_camera(512, 512)
ro, rd = frame_rays(torch.from_numpy(np.ascontiguousarray(pose)).to(dev), intr, 512, 512)
sel = torch.randperm(512 * 512, device=dev, generator=g)[:65536]
ro, rd = ro[sel].contiguous(), rd[sel].contiguous()
aabb = torch.tensor([-1, -0.5, -1, 1, 0.5, 1], dtype=torch.float32, device=dev)
nears, fars = R.near_far_from_aabb(ro, rd, aabb, 0.05)
ctr = torch.zeros(2, dtype=torch.int32, device=dev)
xyz, d, _, rays = R.march_rays_train(ro, rd, 1.0, torch.from_numpy(ones_bitfield()).to(dev), 1, 128, nears, fars, ctr, -1, False, 128, True, 1 / 256, 192,
                                     layout=layout)
M = xyz.shape[0]
x01 = torch.empty(3, M, 2, device=dev)
call("lz_triplane_plane_coords", ptr(xyz.contiguous()), M, 1.0, ptr(x01), stream())
denc = torch.randn(3, 12, M, device=dev, generator=g) * 1e-3
emb = [net.encoder_xy.embeddings, net.encoder_yz.embeddings, net.encoder_xz.embeddings]
ge = torch.zeros((3,) + tuple(emb[0].shape), device=dev)


def run(p):
    call("lz_grid_encode_backward", ptr(denc[p]), ptr(x01[p]), ptr(emb[p]), ptr(net.offsets), ptr(ge[p]), M, 2, 1, 12, net.S, net.H, None, None, 0, 0,
         0, 3, stream())


res = []
for p in range(3):
    for _ in range(3):
        run(p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(p)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 10)
print(f"layout={layout} samples={M} scatter ms per plane: " + " ".join(f"{t:.3f}" for t in res) + f"  sum {sum(res):.3f}", flush=True)

if "--levels" in sys.argv:      # where the time goes: one level at a time (a one-level grid of that level's resolution and table size; timing only)
    off = net.offsets.cpu().numpy()
    for p in (0, 1):
        row = []
        for lv in range(12):
            size = int(off[lv + 1] - off[lv])
            Hl = int(round(64 * 2.0 ** (lv * net.S)))
            o1 = torch.tensor([0, size], dtype=torch.int32, device=dev)
            ge1 = torch.zeros(size, 1, device=dev)
            gl = denc[p, lv].contiguous()
            f = lambda: call("lz_grid_encode_backward", ptr(gl), ptr(x01[p]), ptr(ge1), ptr(o1), ptr(ge1), M, 2, 1, 1, 0.0, Hl, None, None, 0, 0, 0, 3, stream())
            for _ in range(2):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 5)
        print(f"layout={layout} plane {p} per-level us: " + " ".join(f"{1e3 * t:.0f}" for t in row) + f"  sum {1e3 * sum(row):.0f}", flush=True)
