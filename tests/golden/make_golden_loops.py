#!/usr/bin/env python3
"""Generate tests/golden/reference_loops.npz by RUNNING the reference's own render loops and raymarching wrappers (build container only;
SURVEY 8(c) "what the import pins", item 5).  Imported from /root/reference and executed unmodified:

  * NeRFRenderer.render -> run_cuda_for_inference                          nerf_triplane/renderer.py:406-570
  * NeRFRenderer.render -> run_cuda, inference branch (self.evaluate)      renderer.py:304-404
  * NeRFRenderer.run_cuda, training branch                                 renderer.py:279-304, 380-404
  * the wrappers they go through: near_far_from_aabb, march_rays (M += 128 - M % 128, zero-filled rows, torch.rand noise on request),
    march_rays_train (mean_count / force_all_rays sizing, the counter trim), composite_rays_triplane (in place),
    composite_rays_train_triplane                                          raymarching/raymarching.py:18-48, 186-280, 347-398, 594-671
  * NeRFNetwork.forward / encode_audio under them                         nerf_triplane/network.py:226-311

The compiled `_raymarching_face` extension cannot be built here (no nvcc); its ENTRY POINTS are bound to the C checker's kernels
(oracle/raymarch_oracle.c) with the extension's own argument lists (raymarching/src/raymarching.h:7-38), so everything ABOVE the kernels
-- buffer sizing, padding, the n_step schedule, mask compaction, perturb on the first iteration only, abs().sum(-1) ambients, blend / clamp /
depth normalisation -- is the reference's code, and the fixture pins oracle/render.py (the checker's restatement of those loops) to it.
Recorded per run: the image and sums, the (n_alive, n_step, M) of every march_rays call, per-ray marched counts, the noise the wrappers
drew.  Only arrays travel.

Run:  python tests/golden/make_golden_loops.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sys.path for the reference and the checker, encoder back-end adapters)
from make_golden import O  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from frontends_inputs import audio_weights, audio_windows  # noqa: E402
from make_golden_frontends import load_np  # noqa: E402

from lzzx_nerf_amd.synthetic import ellipsoid_grid, make_params, synthetic_camera  # noqa: E402  (numbers only: camera, scene)

H, W = 48, 40
u32, f32c, i32c = C.c_uint32, C.c_float, C.c_int


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Log:
    """what the kernel entry points saw during one reference run"""

    def __init__(self, N):
        self.N = N
        self.calls = []                 # (n_alive, n_step, M) per march_rays
        self.counts = np.zeros(N, np.int64)
        self.noises = None              # first non-zero noise vector the wrappers handed down
        self.acc = None                 # the in-place accumulators of composite_rays_triplane
        self.train = {}


LOG = None


def install_raymarching_kernels():
    """bind _raymarching_face's entry points (raymarching.h:7-38) to the checker's C kernels, argument for argument, in place on the
    torch tensors the reference's wrappers allocated"""
    be = sys.modules["_raymarching_face"]
    L = O.lib()

    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        L.lzo_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32c(min_near), _p(nears), _p(fars))

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, Hh, grid, near, far, xyzs, dirs, deltas, noises):
        assert rays_alive.dtype == torch.int32 and rays_alive.is_contiguous() and xyzs.is_contiguous() and grid.dtype == torch.uint8
        L.lzo_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), f32c(bound), f32c(dt_gamma),
                         u32(max_steps), u32(Cc), u32(Hh), _p(grid), _p(near), _p(far), _p(xyzs), _p(dirs), _p(deltas), _p(noises))
        LOG.calls.append((int(n_alive), int(n_step), int(xyzs.shape[0])))
        valid = (deltas[: n_alive * n_step, 0] != 0).reshape(n_alive, n_step).sum(1).numpy()
        np.add.at(LOG.counts, rays_alive[:n_alive].numpy(), valid)
        if LOG.noises is None and bool((noises != 0).any()):
            LOG.noises = noises.numpy().copy()

    def composite_rays_triplane(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, a0, a1, unc, ws, depth, image, a0s, a1s, us):
        for t in (sigmas, rgbs, deltas, a0, a1, unc):
            assert t.dtype == torch.float32 and t.is_contiguous()
        L.lzo_composite_rays(u32(n_alive), u32(n_step), f32c(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas), _p(rgbs), _p(deltas),
                             _p(a0), _p(a1), _p(unc), i32c(2), i32c(0), i32c(1), _p(ws), _p(depth), _p(image), _p(a0s), _p(a1s), _p(us))
        LOG.acc = dict(weights_sum=ws, depth=depth, image_raw=image, amb_aud_sum=a0s, amb_eye_sum=a1s, uncertainty_sum=us)

    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, Cc, Hh, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
        L.lzo_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), f32c(bound), f32c(dt_gamma), u32(max_steps), u32(N), u32(Cc), u32(Hh), u32(M),
                               _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(rays), _p(counter), _p(noises))
        LOG.train.update(M=int(M), counter=counter.numpy().copy(), noises=noises.numpy().copy())

    def composite_rays_train_triplane_forward(sigmas, rgbs, a0, a1, unc, deltas, rays, M, N, T_thresh, ws, a0s, a1s, us, depth, image):
        for t in (sigmas, rgbs, a0, a1, unc, deltas):
            assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.shape)
        L.lzo_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(a0), _p(a1), _p(unc), _p(deltas), _p(rays), u32(M), u32(N), f32c(T_thresh),
                                           i32c(2), i32c(0), i32c(1), _p(ws), _p(a0s), _p(a1s), _p(us), _p(depth), _p(image))
        LOG.train.update(comp_M=int(M), image_raw=image.numpy().copy(), depth_raw=depth.numpy().copy())

    be.near_far_from_aabb, be.march_rays, be.composite_rays_triplane = near_far_from_aabb, march_rays, composite_rays_triplane
    be.march_rays_train, be.composite_rays_train_triplane_forward = march_rays_train, composite_rays_train_triplane_forward
    torch.Tensor.cuda = lambda self, *a, **k: self   # the wrappers move CPU inputs with .cuda() (raymarching.py:33-34); no GPU here


def ellipsoid_bits():
    inside, coords = ellipsoid_grid()
    grid = np.zeros((1, 128 ** 3), np.float32)
    grid[0, O.morton3D(coords)] = inside.astype(np.float32)
    return O.packbits(grid, 0.5)


def main():
    global LOG
    MG.install_backends()
    install_raymarching_kernels()
    import nerf_triplane.renderer as RR  # reference
    from nerf_triplane.network import NeRFNetwork  # reference
    from nerf_triplane.utils import get_bg_coords, get_rays  # reference
    t = torch.from_numpy
    golden = np.load(os.path.join(HERE, "reference_python.npz"))
    P = make_params(golden)

    torch.manual_seed(0)
    net = NeRFNetwork(MG.Opt())
    sd = net.state_dict()
    for k in ("sigma_net.net.0.weight", "color_net.net.1.weight", "individual_codes"):     # the same seeded network as reference_python.npz
        assert np.array_equal(sd[k].numpy(), golden["sd/" + k]), k
    for n in ("xy", "yz", "xz"):
        getattr(net, f"encoder_{n}").embeddings.data.copy_(t(P[f"encoder_{n}.embeddings"]))
    load_np(net, audio_weights(29))
    net.density_bitfield.copy_(t(ellipsoid_bits()))

    pose, intr = synthetic_camera(H, W)
    rays = get_rays(t(pose[None]), intr, H, W, -1)
    rays_o, rays_d = rays["rays_o"], rays["rays_d"]                      # [1, N, 3]
    N = H * W
    bg_coords = get_bg_coords(H, W, "cpu")
    poses = t(pose[None])
    auds = t(audio_windows(29))
    eye = torch.full((1, 1), 0.25)
    out = dict(rays_o=rays_o[0].numpy(), rays_d=rays_d[0].numpy(), HW=np.array([H, W]), eye=eye.numpy())
    with torch.no_grad():
        net.eval()
        out["enc_a"] = net.encode_audio(auds).numpy()
    out["ind_code"] = net.individual_codes[0].detach().numpy()

    def reset_globals():
        RR.zeroDepth = RR.zero_amb_aud_sum = RR.zero_amb_eye_sum = RR.zero_uncertainty_sum = None     # renderer.py:80-83 caches them forever

    # ---- inference: run_cuda_for_inference and run_cuda's evaluate branch --------------------------------------------------------
    cases = [("ms16", dict(max_steps=16)), ("ms32", dict(max_steps=32)), ("ms64", dict(max_steps=64)),
             ("ms16_T", dict(max_steps=16, T_thresh=0.8)), ("ms32_perturb", dict(max_steps=32, perturb=True)),
             ("ms24_dg0", dict(max_steps=24, dt_gamma=0.0))]
    for tag, kw in cases:
        kw = dict(dict(dt_gamma=1.0 / 256, T_thresh=1e-4, perturb=False), **kw)
        for path in ("inference", "evaluate"):
            reset_globals()
            LOG = Log(N)
            net.eval()
            net.evaluate = path == "evaluate"        # eval_step sets it (TrainerUtil.py:390-394) and leaves `testing` off: unc_net runs
            net.testing = path == "inference"        # test_step sets it around render (TrainerUtil.py:432-436): uncertainty = ln 2
            torch.manual_seed(99)
            with torch.no_grad():
                res, _ = net.render(rays_o, rays_d, auds, bg_coords, poses, eye=eye, index=0, bg_color=None, **kw)
            pre = f"{tag}/{path}/"
            out[pre + "image"] = res["image"][0].numpy().copy()
            out[pre + "schedule"] = np.array(LOG.calls, np.int64).reshape(-1, 3)
            out[pre + "counts"] = LOG.counts.copy()
            for k, v in LOG.acc.items():
                out[pre + k] = v.numpy().copy()
            if LOG.noises is not None:
                out[pre + "noises"] = LOG.noises
            if path == "evaluate":
                out[pre + "depth_norm"] = res["depth"][0].numpy().copy()             # clamp(depth - nears, 0) / (fars - nears), renderer.py:385
                out[pre + "ambient_aud"] = res["ambient_aud"][0].numpy().copy()
                out[pre + "uncertainty"] = res["uncertainty"].numpy().copy()
        assert np.array_equal(out[f"{tag}/inference/image"], out[f"{tag}/evaluate/image"]) or kw["perturb"]
        out[f"{tag}/kw"] = np.array([kw["max_steps"], kw["dt_gamma"], kw["T_thresh"], float(kw["perturb"])], np.float64)
    net.evaluate = net.testing = False

    # ---- training forward: run_cuda, training branch ---------------------------------------------------------------------------------
    tcases = [("t_all", dict(max_steps=32, force_all_rays=True, mean_count=-1)),
              ("t_first", dict(max_steps=32, force_all_rays=False, mean_count=-1)),
              ("t_mean4096", dict(max_steps=32, force_all_rays=False, mean_count=4096)),       # under-estimated: rays dropped (raymarching.cu:457)
              ("t_mean20000_perturb", dict(max_steps=16, force_all_rays=False, mean_count=20000, perturb=True))]
    for tag, kw in tcases:
        kw = dict(dict(perturb=False), **kw)
        reset_globals()
        LOG = Log(N)
        net.train()
        net.mean_count = kw["mean_count"]
        net.local_step = 0
        torch.manual_seed(98)
        with torch.no_grad():
            res, _ = net.run_cuda(rays_o, rays_d, auds, bg_coords, poses, eye=eye, index=0, dt_gamma=1.0 / 256, bg_color=None, perturb=kw["perturb"],
                                  force_all_rays=kw["force_all_rays"], max_steps=kw["max_steps"], T_thresh=1e-4)
        pre = f"{tag}/"
        xyzs = res["rays"][0]
        out[pre + "n_rows"] = np.array([xyzs.shape[0], LOG.train["M"], LOG.train["comp_M"]], np.int64)
        out[pre + "counter"] = LOG.train["counter"]
        out[pre + "noises"] = LOG.train["noises"]
        out[pre + "image"] = res["image"][0].numpy().copy()
        out[pre + "depth_norm"] = res["depth"][0].numpy().copy()
        out[pre + "weights_sum"] = res["weights_sum"].numpy().copy()
        out[pre + "ambient_aud"] = res["ambient_aud"][0].numpy().copy()
        out[pre + "ambient_eye"] = res["ambient_eye"][0].numpy().copy()
        out[pre + "uncertainty"] = res["uncertainty"].numpy().copy()
        out[pre + "xyzs_sum"] = xyzs.numpy().astype(np.float64).sum(0)
        out[pre + "kw"] = np.array([kw["max_steps"], float(kw["force_all_rays"]), kw["mean_count"], float(kw["perturb"])], np.float64)
    net.eval()
    path = os.path.join(HERE, "reference_loops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB;", len(out), "arrays")
    for tag, _ in cases:
        s = out[f"{tag}/inference/schedule"]
        print(tag, "iterations", len(s), "C_eff", int(s[:, 1].sum()), "schedule", [tuple(r) for r in s[:8]], "max count", int(out[f"{tag}/inference/counts"].max()))
    for tag, _ in tcases:
        print(tag, out[f"{tag}/n_rows"], out[f"{tag}/counter"])


if __name__ == "__main__":
    main()
