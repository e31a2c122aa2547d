"""bisect helper: run ONE (D, L, C, T, desired, B) config of the sample-major grid kernel against the level-major one"""
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
from lzzx_nerf_amd._util import call, ptr, stream
from lzzx_nerf_amd.gridencoder import GridEncoder

D, L, C, T, desired, B = [int(v) for v in sys.argv[1:7]]
enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, log2_hashmap_size=T, desired_resolution=desired).cuda()
enc.embeddings.data.uniform_(-1, 1)
S = float(np.float32(np.log2(enc.per_level_scale)))
x = torch.rand(B, D, device="cuda")
a = torch.full((B, L * C), -5.0, device="cuda")
lm = torch.full((L, B, C), -6.0, device="cuda")
call("lz_grid_encode_forward", ptr(x), ptr(enc.embeddings.data), ptr(enc.offsets), ptr(lm), B, D, C, L, S, enc.base_resolution, None, 0, 0, 0, 0, stream())
torch.cuda.synchronize()
call("lz_grid_encode_forward", ptr(x), ptr(enc.embeddings.data), ptr(enc.offsets), ptr(a), B, D, C, L, S, enc.base_resolution, None, 0, 0, 0, 1, stream())
torch.cuda.synchronize()
bad = a != lm.permute(1, 0, 2).reshape(B, L * C)
print("OK", sys.argv[1:7], "mismatch", int(bad.sum()), flush=True)
