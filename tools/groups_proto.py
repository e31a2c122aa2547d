#!/usr/bin/env python3
"""Prototype: render one frame as G independent ray groups, each on its own HIP stream (one host thread per group), so the small
march / composite launches of one group run under the head launch of another."""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_params, synthetic_camera
from lzzx_nerf_amd.head import FusedTriplaneHead
from lzzx_nerf_amd.renderer import TriplaneRenderer, get_rays

device = torch.device("cuda", 0)
golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
P = make_params(golden)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
bits = dev(np.full(128 ** 3 // 8, 255, np.uint8))
H = W = 512
pose, intr = synthetic_camera(H, W)
rays_o, rays_d = get_rays(dev(pose), intr, H, W)
enc_a, ind, eye = dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"])
N = H * W
head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, bound=1.0, device=device)
ref = TriplaneRenderer(head, bits, bound=1.0).render(rays_o, rays_d, enc_a, ind, eye, dt_gamma=1 / 256, max_steps=192, T_thresh=1e-4)["image"].clone()

for G in (1, 2, 3, 4):
    for mode in ("contig", "rows"):
        if G == 1 and mode == "rows":
            continue
        if mode == "contig":
            idx = [torch.arange(N * g // G, N * (g + 1) // G, device=device) for g in range(G)]
        else:   # image rows dealt round-robin: every group sees the same mix of long and short chords
            rows = torch.arange(H, device=device)
            idx = [((rows[rows % G == g])[:, None] * W + torch.arange(W, device=device)[None, :]).reshape(-1) for g in range(G)]
        ro = [rays_o[i].contiguous() for i in idx]
        rd = [rays_d[i].contiguous() for i in idx]
        rs = [TriplaneRenderer(head, bits, bound=1.0) for _ in range(G)]
        streams = [torch.cuda.Stream() for _ in range(G)]
        outs = [None] * G

        def work(g):
            with torch.cuda.stream(streams[g]):
                outs[g] = rs[g].render(ro[g], rd[g], enc_a, ind, eye, dt_gamma=1 / 256, max_steps=192, T_thresh=1e-4)

        def frame():
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
            th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
            for t in th: t.start()
            for t in th: t.join()
            for s in streams:
                cur.wait_stream(s)

        for _ in range(3):
            frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            frame()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        img = torch.empty_like(ref)
        for g in range(G):
            img[idx[g]] = outs[g]["image"]
        samples = sum(int(o["state"][5]) for o in outs)
        print(f"G={G} {mode:6s} {dt * 1e3:7.3f} ms/frame  {samples / dt / 1e9:.3f} Gsamples/s  identical={torch.equal(img, ref)}  iters={[int(o['state'][6]) for o in outs]}", flush=True)
