"""cfg2 hash grid (D3 L16 C2) on RAY-ORDERED sample positions (256 x 256 rays x 128 samples = 2^23 points, ray-major, as
march_rays hands them to the encoder) vs uniformly random points: forward layouts 0 / 1 / 2."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synthetic_camera
from lzzx_nerf_amd.gridencoder import GridEncoder
from lzzx_nerf_amd.renderer import get_rays
from lzzx_nerf_amd._util import call, ptr, stream

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


def timeit(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ray_points(H=256, S=128):
    pose, intr = synthetic_camera(H, H)
    ro, rd = get_rays(torch.from_numpy(pose).to(dev), intr, H, H)
    t = torch.linspace(2.35, 4.35, S, device=dev)                       # through the [-1, 1]^3 box in front of the camera
    p = ro[:, None, :] + rd[:, None, :] * t[None, :, None]               # [rays, S, 3], ray-major
    return ((p.clamp(-1, 1) + 1) / 2).reshape(-1, 3).contiguous()


enc = GridEncoder(desired_resolution=2048).to(dev)
enc.embeddings.data.uniform_(-1, 1, generator=g)
D, C, L = enc.input_dim, enc.level_dim, enc.num_levels
S = float(np.float32(np.log2(enc.per_level_scale)))
B = 1 << 23
for tag, x in (("random", torch.rand(B, D, device=dev, generator=g)), ("ray-ordered", ray_points())):
    assert x.shape[0] == B
    for f16 in (0, 1):
        emb = enc.embeddings.data.half() if f16 else enc.embeddings.data
        out = torch.empty(B, L * C, device=dev, dtype=emb.dtype)
        bps = 12 + 16 * 8 * 2 * (2 if f16 else 4) + (64 if f16 else 128)
        for layout in (0, 1, 2):
            ms = timeit(lambda: call("lz_grid_encode_forward", ptr(x), ptr(emb), ptr(enc.offsets), ptr(out), B, D, C, L, S,
                                     enc.base_resolution, None, 0, 0, f16, layout, stream()))
            print(f"cfg2 {tag:11s} f16={f16} layout={layout}: {ms:.3f} ms  {B / ms / 1e6:.2f} Gsample/s  {bps * B / ms / 1e9:.2f} TB/s algorithmic "
                  f"({bps * B / ms / 1e9 / 8 * 100:.0f} % of 8 TB/s)", flush=True)
