"""the stand-alone triplane plane encoder (D2 L12 C1 f32, the roofline_gridencoder workload): grid_encode on 2^24 uniform samples and on the
cfg3 step's march rows -- tools/plane_bench.py; LZZX_NERF_HIP_SO selects a library variant"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lzzx_nerf_amd.gridencoder import GridEncoder, grid_encode
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
enc = GridEncoder(input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=14, desired_resolution=512).to(dev)
enc.embeddings.data.uniform_(-1, 1, generator=g)
B = 1 << 24
x = torch.rand(B, 2, device=dev, generator=g)
f = lambda: grid_encode(x, enc.embeddings.data, enc.offsets, enc.per_level_scale, enc.base_resolution, False, 0, False)
for _ in range(3):
    f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"plane forward 2^24 random samples: {ms:.4f} ms, {248 * B / ms / 1e6:.1f} GB/s algorithmic ({248 * B / ms / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
