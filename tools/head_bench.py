"""fused-head launch time vs row count (f32 and f16): separates the per-launch overhead from the per-row cost"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_params
from lzzx_nerf_amd.head import FusedTriplaneHead

golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
P = make_params(golden)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
Mmax = 1 << 21
xyz = torch.rand(Mmax, 3, device=dev, generator=g) * 2 - 1
d = torch.nn.functional.normalize(torch.randn(Mmax, 3, device=dev, generator=g), dim=-1)
enc_a, ind, eye = [torch.from_numpy(golden[k]).to(dev) for k in ("net_enc_a", "net_ind", "net_eye")]
for prec in ("f32", "f16"):
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, precision=prec)
    out = tuple(torch.empty(s, device=dev) for s in ((Mmax,), (Mmax, 3), (Mmax, 1), (Mmax, 1), (Mmax, 1)))
    for M in (1 << 21, 1 << 20, 1 << 19, 262144, 200000, 131072, 65536, 32768, 16384, 4096, 256):
        f = lambda: head.forward(xyz[:M], d[:M], enc_a, ind, eye, out=tuple(o[:M] for o in out))
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"{prec} rows {M:8d}: {us:9.2f} us/launch  {us / M * 1e3:8.3f} ns/row", flush=True)

# launch floor: count = 0 (kernel returns before staging its weights) vs one slice
head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, precision="f32")
out = tuple(torch.empty(s, device=dev) for s in ((4096,), (4096, 3), (4096, 1), (4096, 1), (4096, 1)))
for cnt in (0, 16, 4096):
    c = torch.tensor([cnt], dtype=torch.int32, device=dev)
    f = lambda: head.forward(xyz[:4096], d[:4096], enc_a, ind, eye, count_ptr=c.data_ptr(), out=out)
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"f32 count={cnt}: {e0.elapsed_time(e1) / 50 * 1e3:.2f} us/launch (includes the Python call)")
