"""micro-benchmark of the grid encoder kernels (forward layouts 0/1/2, backward) on the cfg2 and triplane shapes"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lzzx_nerf_amd.gridencoder import GridEncoder
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


cases = [("cfg2", dict(desired_resolution=2048), 1 << 23), ("triplane", dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=64,
                                                                           log2_hashmap_size=14, desired_resolution=512), 1 << 22)]
for tag, kw, B in cases:
    enc = GridEncoder(**kw).to(dev)
    enc.embeddings.data.uniform_(-1, 1, generator=g)
    D, C, L = enc.input_dim, enc.level_dim, enc.num_levels
    S = float(np.float32(np.log2(enc.per_level_scale)))
    x = torch.rand(B, D, device=dev, generator=g)
    for f16 in (0, 1):
        if f16 and C % 2:
            continue
        emb = enc.embeddings.data.half() if f16 else enc.embeddings.data
        out = torch.empty(B, L * C, device=dev, dtype=emb.dtype)
        for layout in (0, 1, 2):
            ms = timeit(lambda: call("lz_grid_encode_forward", ptr(x), ptr(emb), ptr(enc.offsets), ptr(out), B, D, C, L, S,
                                     enc.base_resolution, None, 0, 0, f16, layout, stream()))
            print(f"{tag} fwd f16={f16} layout={layout}: {ms:.3f} ms  ({B / ms / 1e6:.2f} Gsample/s)", flush=True)
    emb = enc.embeddings.data
    grad = torch.rand(B, L * C, device=dev, generator=g)
    gemb = torch.zeros_like(emb)
    for layout in (0, 1, 2):
        ms = timeit(lambda: call("lz_grid_encode_backward", ptr(grad), ptr(x), ptr(emb), ptr(enc.offsets), ptr(gemb), B, D, C, L, S,
                                 enc.base_resolution, None, None, 0, 0, 0, layout, stream()))
        print(f"{tag} bwd f32 grad_layout={layout}: {ms:.3f} ms  ({B / ms / 1e6:.2f} Gsample/s)", flush=True)
