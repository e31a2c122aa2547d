"""the head kernels of the -O training step one by one on M random samples (default 5.95 M, the cfg3 step's count): inference f16 head,
recording forward, light forward (recompute arrangement), and the two fused backwards -- torch events around 10 launches each"""
import ctypes as C
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_params
from lzzx_nerf_amd.head import FusedTriplaneHead
from lzzx_nerf_amd.head_train import FusedTriplaneTrainHead

golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
P = make_params(golden)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
# the cfg3 step's samples: 65 536 random rays of the 512 x 512 frame marched through the all-ones grid (ray-major: a ray's samples are
# consecutive rows, as in the step; uniformly random points would make every table read its own cache line)
from lzzx_nerf_amd import raymarching as R
from lzzx_nerf_amd.synthetic import ones_bitfield, synthetic_camera
from lzzx_nerf_amd.utils import frame_rays
pose, intr = synthetic_camera(512, 512)
ro, rd = frame_rays(torch.from_numpy(np.ascontiguousarray(pose)).to(dev), intr, 512, 512)
sel = torch.randperm(512 * 512, device=dev, generator=g)[:65536]
if "--sorted" in sys.argv:      # the same random rays presented in pixel order
    sel = sel.sort().values
    sys.argv.remove("--sorted")
ro, rd = ro[sel].contiguous(), rd[sel].contiguous()
aabb = torch.tensor([-1, -0.5, -1, 1, 0.5, 1], dtype=torch.float32, device=dev)
nears, fars = R.near_far_from_aabb(ro, rd, aabb, 0.05)
ctr = torch.zeros(2, dtype=torch.int32, device=dev)
xyz, d, _, rays = R.march_rays_train(ro, rd, 1.0, torch.from_numpy(ones_bitfield()).to(dev), 1, 128, nears, fars, ctr, -1, False, 128, True, 1 / 256, 192)
xyz, d = xyz.detach().contiguous(), d.detach().contiguous()
M = xyz.shape[0]


def interleave_perm(rays, M, G):
    """processing order that walks groups of G consecutive rays step by step (step k of every ray of the group, then step k + 1 ...):
    perm[p] = buffer row of processing position p; rows past the last sample keep their place"""
    rays = rays.long()
    off, cnt = rays[:, 1], rays[:, 2]
    N = rays.shape[0]
    total = int((off + cnt).max())
    ray_of = torch.repeat_interleave(torch.arange(N, device=dev), cnt)            # rows are ray-major in ray order
    assert ray_of.numel() == total and bool((off[1:] == (off + cnt)[:-1]).all())
    k = torch.arange(total, device=dev) - off[ray_of]
    g0 = (ray_of // G) * G
    gbase = off[g0]
    pos = gbase.clone()
    for j in range(G):
        rj = torch.clamp(g0 + j, max=N - 1)
        cj = torch.where(g0 + j < N, cnt[rj], torch.zeros_like(cnt[rj]))
        pos += torch.minimum(cj, k) + ((cj > k) & (g0 + j < ray_of)).long()
    perm = torch.arange(M, device=dev)
    perm[pos] = torch.arange(total, device=dev)
    assert bool((perm.sort().values == torch.arange(M, device=dev)).all())
    return perm


for a in list(sys.argv):
    if a.startswith("--interleave="):
        G = int(a.split("=")[1])
        sys.argv.remove(a)
        perm = interleave_perm(rays, M, G)
        xyz, d = xyz[perm].contiguous(), d[perm].contiguous()
        print("interleaved over groups of", G, "rays", flush=True)
print("samples", M, flush=True)
enc_a, ind, eye = [torch.from_numpy(golden[k]).to(dev) for k in ("net_enc_a", "net_ind", "net_eye")]


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


head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, precision="f16")
out = tuple(torch.empty(s, device=dev) for s in ((M,), (M, 3), (M, 1), (M, 1), (M, 1)))
print(f"inference f16 head (32-sample slices): {timeit(lambda: head.forward(xyz, d, enc_a, ind, eye, out=out)):.3f} ms", flush=True)
gout = [torch.randn(s, device=dev, generator=g) * 1e-3 for s in ((M,), (M, 3), (M, 1), (M, 1), (M, 1))]
for rc in (False, True):
    net = FusedTriplaneTrainHead(P, bound=1.0, forward_dtype="f16", backward_dtype="f16", recompute_mlp=rc).to(dev)
    outs = [None]

    def fwd():
        outs[0] = net(xyz, d, enc_a, ind, eye)

    def fwd_bwd():
        o = net(xyz, d, enc_a, ind, eye)
        torch.autograd.backward(list(o), gout)
    with torch.no_grad():
        pass
    t_f = timeit(fwd)
    t_fb = timeit(fwd_bwd)
    print(f"recompute_mlp={rc}: forward {t_f:.3f} ms, forward + backward (with table scatters, torch glue) {t_fb:.3f} ms", flush=True)
