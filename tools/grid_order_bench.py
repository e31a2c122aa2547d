#!/usr/bin/env python3
"""ONE ordered case of bench.py's roofline_gridencoder per process (so that a PMC pass can tell them apart: they share kernel names):
tools/grid_order_bench.py ray_f32 | march_f32 | ray_f16 | march_f16 -- the cfg2 hash grid (D3 L16 C2) forward on 2^23 sample positions in
march_rays_train order (256 x 256 rays x 128 consecutive samples, ray-major) or in the inference loop's order ([iteration][ray][8 steps])."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lzzx_nerf_amd.gridencoder import GridEncoder, grid_encode
from lzzx_nerf_amd.synthetic import synthetic_camera
from lzzx_nerf_amd.utils import frame_rays

case = sys.argv[1]
order, prec = case.split("_")
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
enc = GridEncoder(desired_resolution=2048).to(dev)
enc.embeddings.data.uniform_(-1, 1, generator=g)
pose, intr = synthetic_camera(256, 256)
ro, rd = frame_rays(torch.from_numpy(pose).to(dev), intr, 256, 256)
if order == "march":
    t = torch.linspace(2.35, 4.35, 128, device=dev).view(16, 1, 8)
    p = ro[None, :, None, :] + rd[None, :, None, :] * t[..., None]
else:
    t = torch.linspace(2.35, 4.35, 128, device=dev)
    p = ro[:, None, :] + rd[:, None, :] * t[None, :, None]
x = ((p.clamp(-1, 1) + 1) / 2).reshape(-1, 3).contiguous()
emb = enc.embeddings.data.half() if prec == "f16" else enc.embeddings.data
for _ in range(6):
    grid_encode(x, emb, enc.offsets, enc.per_level_scale, enc.base_resolution, False, 0, False)
torch.cuda.synchronize()
print(case, x.shape[0])
