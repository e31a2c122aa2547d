#!/usr/bin/env python3
"""Headline benchmark: 512x512 triplane-head inference render at max_steps = 192 on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" renders one full synthetic frame per rank through the whole hot path (near/far -> [march -> fused
head -> composite -> compaction] until every ray is done -> blend), inputs resident in HBM, no host sync inside
the frame.  Workload: camera at (0,0,-3.35) looking down +z, fovy 21.24 deg, bound 1, aabb
[-1,-.5,-1,1,.5,1], dt_gamma 1/256, T_thresh 1e-4, all-ones occupancy grid (dense: nothing is skipped), triplane
tables U(-1,1), MLP weights = the reference's torch init under seed 0 (tests/golden fixture), enc_a ~ N(0,1),
eye 0.25.  Data is synthetic.

N > 1 (weak scaling): the global batch is N consecutive frames of a talking-head clip -- what the reference renders
(TrainerUtil.test): the head pose sways a little from frame to frame (0.01 rad per frame about the vertical axis) and every
frame has its own audio feature -- ray-sharded contiguously, i.e. rank r renders frame r; every step ends with ONE RCCL
all-gather of the rendered RGB tiles so that every rank holds the whole batch.  Frame 0 (N = 1) is the frontal pose with the
fixture's audio feature.

Iteration schedule: every loop iteration marches n_step = max(min(F * N // n_alive, C), 1) samples per alive ray.  The reference uses
F = 1, C = 8 (renderer.py:513), i.e. N sample rows per iteration and thin launches while most rays are alive; pixels do not depend on F
and C (each ray marches the same sample sequence and compositing resumes exactly), so the headline runs F = C = 4 (4 N rows
per iteration, 29 instead of 114 iterations for this frame) and the `reference_schedule` leg times F = 1, C = 8 on the same frame and
checks that the image and the sample count are identical.

Prints one JSON line on rank 0.  `value` = marched samples (delta != 0) per second over all ranks.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_SAMPLE = 46368          # 23 184 MAC per sample, inference head
ISSUED_FLOP_PER_ROW = 361 * 2048 // 16   # the head issues 361 v_mfma_f32_16x16x4_f32 (2048 FLOP each) per 16-row slice
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, v_mfma_f32_16x16x4_f32
HBM_PEAK_GBS = 8000.0


def orbit_pose(k, n):
    """head pose of frame k of an n-frame clip: camera on a circle of radius 3.35 about the vertical axis, 0.01 rad per frame, frame 0 frontal"""
    th = 0.01 * k
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], dtype=np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R
    pose[:3, 3] = R @ np.array([0, 0, -3.35], dtype=np.float32)
    return pose


# kernels behind each grid_roofline case and the batch size tools/grid_bench.py profiled them at (tools/profile_grid.sh)
_GRID_PMC = {"triplane_plane_D2_L12_C1_f32": (["lz_k_grid_forward_lds<float, 2u, 1u>"], 1 << 22),
             "hashgrid_D3_L16_C2_f32": (["lz_k_grid_forward_lmp<float, 3u, 2u>", "lz_k_grid_untile"], 1 << 23),
             "hashgrid_D3_L16_C2_f16": (["lz_k_grid_forward_lmp<__half, 3u, 2u>", "lz_k_grid_untile"], 1 << 23),
             "triplane_plane_D2_L12_C1_f32_backward": (["lz_k_grid_backward_lds_fx<2u, 1u>"], 1 << 22)}


def _grid_traffic(tag, B):
    """HBM-side bytes per launch at batch size B from the committed PMC summary (KiB per launch at the profiled batch size, scaled
    per sample; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 -- calibrated there for wide streaming reads, so an
    upper bound for the gather-dominated kernels).  None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r1_grid_pmc_summary.json")
    if not os.path.exists(path):
        return None
    try:
        pmc = json.load(open(path))
        kernels, b_prof = _GRID_PMC[tag]
        kib = 0.0
        for k in kernels:
            f, w = pmc["FETCH_SIZE"][k], pmc["WRITE_SIZE"][k]
            # lz_k_grid_untile is shared by several cases of the profiled script: take its largest launch (the f32 cfg2 one) for f32,
            # half of it for f16
            if k == "lz_k_grid_untile":
                scale = 0.5 if "f16" in tag else 1.0
                kib += (2 * f["max"] + w["max"]) * scale
            else:
                kib += 2 * f["avg_per_launch"] + w["avg_per_launch"]
        return round(kib * 1024 / b_prof * B)
    except (KeyError, ValueError):
        return None


def grid_roofline(device):
    """stand-alone grid encoder: algorithmic bytes / event-timed launch duration"""
    from lzzx_nerf_amd.gridencoder import GridEncoder, grid_encode
    res = {}
    g = torch.Generator(device=device).manual_seed(0)
    tri = dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=14, desired_resolution=512)
    for tag, kw, bytes_per_sample, B, mode in (
            ("triplane_plane_D2_L12_C1_f32", tri, 8 + 12 * 4 * 4 + 48, 1 << 24, "fwd"),
            ("hashgrid_D3_L16_C2_f32", dict(desired_resolution=2048), 12 + 16 * 8 * 2 * 4 + 128, 1 << 23, "fwd"),
            ("hashgrid_D3_L16_C2_f32_ray_ordered", dict(desired_resolution=2048), 12 + 16 * 8 * 2 * 4 + 128, 1 << 23, "fwd_rays"),
            ("hashgrid_D3_L16_C2_f16", dict(desired_resolution=2048), 12 + 16 * 8 * 2 * 2 + 64, 1 << 23, "fwd16"),
            ("triplane_plane_D2_L12_C1_f32_backward", tri, 8 + 48 + 12 * 4 * 4 * 2, 1 << 22, "bwd")):
        enc = GridEncoder(**kw).to(device)
        enc.embeddings.data.uniform_(-1, 1, generator=g)
        if mode == "fwd_rays":   # BASELINE cfg2 as march_rays hands it to the encoder: 256 x 256 rays x 128 samples, ray-major
            from conftest import synthetic_camera
            from lzzx_nerf_amd.renderer import get_rays
            pose, intr = synthetic_camera(256, 256)
            ro, rd = get_rays(torch.from_numpy(pose).to(device), intr, 256, 256)
            t = torch.linspace(2.35, 4.35, 128, device=device)
            x = (((ro[:, None, :] + rd[:, None, :] * t[None, :, None]).clamp(-1, 1) + 1) / 2).reshape(-1, 3).contiguous()
        else:
            x = torch.rand(B, enc.input_dim, device=device, generator=g)
        emb = enc.embeddings.data.half() if mode == "fwd16" else enc.embeddings.data
        if mode == "bwd":
            from lzzx_nerf_amd._util import call, ptr, stream
            grad = torch.rand(B, enc.output_dim, device=device, generator=g)
            gemb = torch.zeros_like(emb)
            S = float(np.float32(np.log2(enc.per_level_scale)))
            f = lambda: call("lz_grid_encode_backward", ptr(grad), ptr(x), ptr(emb), ptr(enc.offsets), ptr(gemb), B, enc.input_dim,
                             enc.level_dim, enc.num_levels, S, enc.base_resolution, None, None, 0, 0, 0, 2, stream())
        else:
            f = lambda: grid_encode(x, emb, enc.offsets, enc.per_level_scale, enc.base_resolution, False, 0, False)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        gbs = bytes_per_sample * B / (ms * 1e-3) / 1e9
        res[tag] = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                        traffic=_grid_traffic(tag, B), samples=B, ms=round(ms, 4), bytes_per_sample=bytes_per_sample)
        del enc, x
    return res


class TriplaneTrainNet(torch.nn.Module):
    """caller-side graph of the reference's NeRFNetwork.forward in training mode (network.py:252-311) on the operator API:
    3 GridEncoders + SHEncoder from get_encoder(), bias-free torch Linear stacks (rocBLAS), autograd"""

    def __init__(self, P, device, mlp="lz"):
        super().__init__()
        from lzzx_nerf_amd.encoding import get_encoder
        from lzzx_nerf_amd.linear import lz_linear
        self.use_lz, self.lz_linear = mlp == "lz", lz_linear
        mk = lambda: get_encoder("hashgrid", input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=14,
                                 desired_resolution=512)[0]
        self.encoder_xy, self.encoder_yz, self.encoder_xz = mk(), mk(), mk()
        self.encoder_dir = get_encoder("spherical_harmonics")[0]
        self.W = torch.nn.ParameterDict({k.replace(".", "_"): torch.nn.Parameter(torch.from_numpy(v).clone()) for k, v in P.items()
                                         if k.endswith(".weight")})
        self.to(device)
        with torch.no_grad():
            for n in ("xy", "yz", "xz"):
                getattr(self, "encoder_" + n).embeddings.copy_(torch.from_numpy(P[f"encoder_{n}.embeddings"]))

    def mlp(self, h, name, n):
        for i in range(n):
            if self.use_lz:   # csrc/lz_linear.hip: one MFMA kernel per layer forward (ReLU fused), two backward
                h = self.lz_linear(h, self.W[f"{name}_net_{i}_weight"], i < n - 1)
            else:
                h = torch.nn.functional.linear(h, self.W[f"{name}_net_{i}_weight"])
                if i < n - 1:
                    h = torch.relu(h)
        return h

    def forward(self, x, d, enc_a, ind, eye):
        enc_x = torch.cat([self.encoder_xy(x[:, :2], bound=1), self.encoder_yz(x[:, 1:], bound=1), self.encoder_xz(x[:, [0, 2]], bound=1)], -1)
        att = self.mlp(enc_x, "aud_ch_att_net", 2)
        eye_att = torch.sigmoid(self.mlp(enc_x, "eye_att_net", 2))
        h = self.mlp(torch.cat([enc_x, enc_a * att, eye * eye_att], -1), "sigma_net", 3)
        sigma = torch.exp(h[:, 0])
        hc = torch.cat([self.encoder_dir(d), h[:, 1:], ind.expand(x.shape[0], -1)], -1)
        rgb = torch.sigmoid(self.mlp(hc, "color_net", 2)) * 1.002 - 0.001
        unc = torch.nn.functional.softplus(self.mlp(enc_x.detach(), "unc_net", 2))[:, 0]
        return sigma, rgb, att.norm(dim=-1), eye_att.abs().sum(-1), unc


def train_bench(args, device, P, golden, bits, rank=0, world=1):
    """BASELINE cfg3: one training step (fwd + bwd + Adam) on `--train-rays` random rays of the 512x512 frame through the
    operator API as the reference's run_cuda arranges it (renderer.py:279-304).  The MLP GEMMs are torch/rocBLAS here;
    everything else is this repo's HIP kernels.  Reported beside the headline line, never instead of it."""
    from conftest import synthetic_camera
    from lzzx_nerf_amd import dist as D
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd.renderer import get_rays
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    H = W = args.size
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(dev(pose), intr, H, W)
    g = torch.Generator(device=device).manual_seed(0)
    n_rays = min(args.train_rays, H * W)
    sel = torch.randperm(H * W, device=device, generator=g)[:n_rays]
    target = torch.rand(n_rays, 3, device=device, generator=g)
    lo, hi = D.shard_bounds(n_rays, rank, world)   # data parallel over the sampled rays (world == 1: everything)
    ro, rd, target = ro[sel[lo:hi]].contiguous(), rd[sel[lo:hi]].contiguous(), target[lo:hi].contiguous()
    if args.train_mlp == "fused":   # one kernel forward, one kernel for the backward data chain (lzzx_nerf_amd/head_train.py)
        from lzzx_nerf_amd.head_train import FusedTriplaneTrainHead
        net = FusedTriplaneTrainHead(P, bound=1.0).to(device)
    else:
        net = TriplaneTrainNet(P, device, mlp=args.train_mlp)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, eps=1e-15, fused=True)
    bucket = D.GradientBucket(net.parameters()) if world > 1 else None   # every .grad a view of one flat buffer: ONE all-reduce per step
    enc_a, ind, eye = dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"])
    aabb = dev(np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32))
    bitfield = dev(bits)
    ctr = torch.zeros(2, dtype=torch.int32, device=device)
    n_samples = [0]

    def step():
        nears, fars = R.near_far_from_aabb(ro, rd, aabb, 0.05)
        ctr.zero_()
        xyzs, dirs, deltas, rays = R.march_rays_train(ro, rd, 1.0, bitfield, 1, 128, nears, fars, ctr, -1, True, 128, True, 1 / 256,
                                                      args.max_steps)
        sigma, rgb, a0, a1, unc = net(xyzs, dirs, enc_a, ind, eye)
        if args.train_mlp == "fused":
            a0, a1, unc = a0[:, 0], a1[:, 0], unc[:, 0]
        ws, a0s, a1s, us, dep, img = R.composite_rays_train_triplane(sigma, rgb, a0, a1, unc, deltas, rays)
        loss = ((img + (1 - ws).unsqueeze(-1) - target) ** 2).mean() + 1e-4 * a0s.mean() + 1e-4 * a1s.mean() + 1e-3 * us.mean()
        if bucket is None:
            opt.zero_grad(set_to_none=True)
            loss.backward()
        else:
            bucket.zero()
            loss.backward()
            bucket.all_reduce()
        opt.step()
        n_samples[0] = xyzs.shape[0]
        return loss

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if world > 1:
        red = torch.tensor([dt, float(n_samples[0])], dtype=torch.float64, device=device)
        mx = red.clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(red)
        dt, n_samples[0] = float(mx[0]), int(red[1])
    return dict(parallelism=f"data parallel over ray shards x{world}, one all-reduce of the flat gradient buffer "
                            f"({bucket.flat.numel() * 4 / 1e6:.2f} MB) per step" if world > 1 else "single GPU",
                workload=f"cfg3: {n_rays} random rays of a {H}x{W} frame, max_steps {args.max_steps}, occupancy={args.scene}, "
                         "march_rays_train -> head (see 'mlp') -> composite_rays_train_triplane -> MSE -> backward (weight gradients + grid "
                         "scatter-add) -> Adam", rays=n_rays, samples_per_step=int(n_samples[0]),
                ms_per_step=round(dt * 1e3, 3), samples_per_s=round(n_samples[0] / dt, 1), rays_per_s=round(n_rays / dt, 1),
                loss=float(loss.detach()), dtype="f32", mlp={"fused": "fused head forward + backward kernels (csrc/lz_head.hip, lz_head_bwd.hip)", "lz": "csrc/lz_linear.hip (MFMA f32)",
                     "torch": "torch/rocBLAS"}[args.train_mlp])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget-factor", type=int, default=4,
                    help="sample rows per iteration = factor x rays (reference: 1; pixels and sample counts do not depend on it)")
    ap.add_argument("--n-step-cap", type=int, default=4, help="max samples per ray per iteration (reference: 8)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"],
                    help="f16 = the reference's opt.fp16 / autocast arithmetic on the f16 matrix cores (not bit-exact vs the f32 checker)")
    ap.add_argument("--train", action="store_true", help="(default now) time a cfg3 training step and add it as 'train_step'")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--train-rays", type=int, default=65536)
    ap.add_argument("--train-dp", action="store_true",
                    help="N > 1 only: also time the cfg3 training step data-parallel over ray shards (strong scaling of one step, "
                         "one gradient all-reduce per step); off by default so the scaling run measures the headline path alone")
    ap.add_argument("--train-mlp", default="fused", choices=["fused", "lz", "torch"],
                    help="training step: fused head forward/backward kernels, per-layer lz_linear kernels, or torch Linear")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--max-steps", type=int, default=192)
    ap.add_argument("--scene", default="ones", choices=["ones", "ellipsoid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip the 8 extra 2M-row head launches that measure the sustained shader clock (tools/profile_bench.sh: the kernel "
                         "trace then holds only warm-up + timed frames, so its average head duration is directly bench's avg_launch_ms_all)")
    ap.add_argument("--no-grid-roofline", action="store_true")
    ap.add_argument("--no-reference-schedule", "--no-fat-schedule", dest="no_fat_schedule", action="store_true",
                    help="skip the leg that re-renders the frame under the reference's own iteration schedule (1 x N rows, <= 8 steps)")
    ap.add_argument("--no-fp16-leg", action="store_true")
    ap.add_argument("--no-occupancy", action="store_true")
    ap.add_argument("--no-dense192", action="store_true")
    ap.add_argument("--gather", default="f32", choices=["f32", "rgb24"],
                    help="what the per-step all-gather moves: f32 RGB tiles, or the video pipe's RGB24 quantised on device (4x fewer bytes)")
    args = ap.parse_args()

    from lzzx_nerf_amd import _lib, dist as D
    rank, world = D.init_from_env()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if world == 1:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())
    assert _lib.load().lz_device_ok() == 1, "bench needs a gfx950 device; there is no fallback path"

    from conftest import ellipsoid_bitfield, make_params, synthetic_camera
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer, get_rays

    golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
    P = make_params(golden)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, bound=1.0, device=device, precision=args.precision)
    bits = np.full(128 ** 3 // 8, 255, np.uint8) if args.scene == "ones" else ellipsoid_bitfield()[0]
    renderer = TriplaneRenderer(head, dev(bits), bound=1.0, budget_factor=args.budget_factor, n_step_cap=args.n_step_cap)
    H = W = args.size
    _, intr = synthetic_camera(H, W)
    pose = orbit_pose(rank, world)   # frame `rank` of the clip
    rays_o, rays_d = get_rays(dev(pose), intr, H, W)   # this rank's shard of the global ray batch = frame `rank`
    enc_a, ind, eye = dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"])
    if rank > 0:   # every frame of the clip has its own audio feature (frame 0: the fixture's)
        enc_a = enc_a + 0.5 * torch.randn(enc_a.shape, device=device, generator=torch.Generator(device=device).manual_seed(100 + rank))
    N = H * W

    def step():
        out = renderer.render(rays_o, rays_d, enc_a, ind, eye, dt_gamma=1 / 256, max_steps=args.max_steps, T_thresh=1e-4,
                              rgb24=args.gather == "rgb24")
        tile = out["image_rgb24"] if args.gather == "rgb24" else out["image"]
        tiles = D.gather_tiles(tile) if world > 1 else tile
        return out, tiles

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    renderer.timing_start(args.steps * args.max_steps + 16)   # HIP event pair around every head launch, on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, tiles = step()
    barrier()
    dt = time.perf_counter() - t0
    head_ms = renderer.timing_stop()
    # sustained shader clock under the head's load: 8 back-to-back 2M-row launches, counters of one wave of the last one
    probe = (C.c_uint64 * 2)()
    if not args.no_clock_probe:
        gp = torch.Generator(device=device).manual_seed(1)
        xs = torch.rand(1 << 21, 3, device=device, generator=gp) * 2 - 1
        ds = torch.nn.functional.normalize(torch.randn(1 << 21, 3, device=device, generator=gp), dim=-1)
        for _ in range(8):
            head.forward(xs, ds, enc_a, ind, eye)
        _lib.call("lz_debug_head_clocks", probe)
        del xs, ds
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    state = out["state"].cpu().numpy()
    samples_per_frame = int(state[5])
    iters_per_frame = int(state[6])
    rows_per_frame = int(state[72])   # LZ_LOOP_STAT_ROWS: rows the head evaluated (n_alive * n_step, exhausted rows included)
    total_samples = torch.tensor([samples_per_frame], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(total_samples)
    total_samples = float(total_samples.item())  # one frame per rank per step
    value = total_samples * args.steps / dt
    rays_per_s = N * world * args.steps / dt

    train_dp = None
    if world > 1 and args.train_dp:   # every rank takes part: one gradient all-reduce per step
        train_dp = train_bench(args, device, P, golden, bits, rank, world)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    # ---- roofline of the dominant kernel (fused head, MFMA-bound): live HIP events from the timed steps ----
    head_total_ms = float(np.sum(head_ms))
    n_launch = len(head_ms)
    launches_with_work = iters_per_frame * args.steps
    achieved_tflops = FLOP_PER_SAMPLE * samples_per_frame * args.steps / (head_total_ms * 1e-3) / 1e12
    roofline = dict(bound="mfma", achieved=round(achieved_tflops, 3), peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(achieved_tflops / F32_MFMA_PEAK_TFLOPS, 4), traffic=None, kernel="lz_k_triplane_head<false>",
                    avg_launch_ms=round(head_total_ms / max(launches_with_work, 1), 5),
                    avg_launch_ms_all=round(head_total_ms / max(n_launch, 1), 5), launches=n_launch,
                    launches_with_work=launches_with_work, flop_per_sample=FLOP_PER_SAMPLE,
                    head_time_share=round(head_total_ms * 1e-3 / dt, 4), rows_per_frame=rows_per_frame,
                    shader_clock_mhz_under_load=round(probe[0] / max(probe[1], 1) * 100.0, 1) if (args.precision == "f32" and probe[1]) else None,
                    issued_frac=round(ISSUED_FLOP_PER_ROW * rows_per_frame * args.steps / (head_total_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4))
    if args.precision == "f16":   # 59 v_mfma_f32_16x16x32_f16 per slice: priced against the dense f16 peak; really gather-rate bound
        roofline.update(peak=2500.0, frac=round(achieved_tflops / 2500.0, 5), kernel="lz_k_triplane_head_f16",
                        issued_frac=round(59 * 16384 / 16 * rows_per_frame * args.steps / (head_total_ms * 1e-3) / 1e12 / 2500.0, 5),
                        note="matrix work is 7% of the f32 kernel's; the kernel is bound by the 144 table gathers per sample")
    # HBM-side traffic of the head per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs, tools/profile_bench.sh): KiB per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    # for gfx950 (128-B requests tallied at 64 B).  bench.py cannot collect counters itself; null when the summary is absent.
    pmc_path = os.path.join(ROOT, "profiles", "r1_final_pmc_summary.json")
    if os.path.exists(pmc_path) and args.precision == "f32":
        try:
            pmc = json.load(open(pmc_path))
            k = "lz_k_triplane_head<false>"
            roofline["traffic"] = round((2 * pmc["FETCH_SIZE"][k]["avg_per_launch"] + pmc["WRITE_SIZE"][k]["avg_per_launch"]) * 1024)
            roofline["traffic_unit"] = "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, averaged over all launches of profiles/r1_final_pmc_summary.json)"
            roofline["algorithmic_bytes_per_launch"] = round(52 * rows_per_frame * args.steps / max(n_launch, 1))
        except (KeyError, ValueError):
            pass
    result = {
        "metric": f"rendered samples/s ({H}x{W} triplane head, max_steps {args.max_steps})", "value": round(value, 1), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "f16 (f32 accumulate, torch-autocast rounding)", "data": "synthetic",
        "config": {"workload": f"{H}x{W} inference frame per GPU, max_steps {args.max_steps}, triplane head (3x D2/L12/C1 hash grid + "
                               f"audio/eye cond + SH4), occupancy={args.scene}, bound 1, dt_gamma 1/256, T_thresh 1e-4",
                   "rays_per_gpu": N, "samples_per_frame": samples_per_frame, "iterations_per_frame": iters_per_frame,
                   "nominal_samples_per_frame": N * args.max_steps, "parallelism": f"ray-sharded x{world}, 1 all-gather/step ({args.gather} tiles)",
                   "schedule": f"n_step = max(min({args.budget_factor} * N // n_alive, {args.n_step_cap}), 1)"
                               + (" (the reference's, renderer.py:513)" if (args.budget_factor, args.n_step_cap) == (1, 8) else
                                  " -- sample rows per iteration sized for 288 GB of HBM; the reference's rule is 1 x N rows, <= 8 steps "
                                  "(renderer.py:513): same pixels and per-ray sample counts, timed in 'reference_schedule'")},
        "rays_per_s": round(rays_per_s, 1),
        "samples_per_ray_mean": round(samples_per_frame / N, 2),
        "roofline": roofline,
    }
    def side_leg(h, budget_factor, n_step_cap):
        """same frame, K timed steps after 2 warm-ups, with another head precision and / or iteration schedule"""
        r2 = TriplaneRenderer(h, dev(bits), bound=1.0, budget_factor=budget_factor, n_step_cap=n_step_cap)
        rr = lambda: r2.render(rays_o, rays_d, enc_a, ind, eye, dt_gamma=1 / 256, max_steps=args.max_steps, T_thresh=1e-4)
        for _ in range(2):
            rr()
        torch.cuda.synchronize()
        r2.timing_start(args.steps * args.max_steps + 16)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o2 = rr()
        torch.cuda.synchronize()
        d2 = time.perf_counter() - t0
        hms = float(np.sum(r2.timing_stop()))
        s2 = o2["state"].cpu().numpy()
        img2 = o2["image"].clone()
        return dict(schedule=f"n_step = max(min({budget_factor} * N // n_alive, {n_step_cap}), 1)", value=round(int(s2[5]) * args.steps / d2, 1),
                    unit="samples/s", ms_per_step=round(d2 / args.steps * 1e3, 4), iterations_per_frame=int(s2[6]),
                    rows_per_frame=int(s2[72]), head_ms_per_step=round(hms / args.steps, 4)), img2, hms, s2

    REF_SCHEDULE = (1, 8)   # renderer.py:513
    if world == 1 and (args.budget_factor, args.n_step_cap) != REF_SCHEDULE and not args.no_fat_schedule:
        try:
            # the same frame under the reference's own iteration schedule (1 x N sample rows per iteration, <= 8 steps per ray): more,
            # thinner launches; pixels and per-ray sample counts must be identical
            leg, fimg, fms, fst = side_leg(head, *REF_SCHEDULE)
            leg["schedule"] += " (the reference's, renderer.py:513)"
            leg["image_equal_to_headline_schedule"] = bool(torch.equal(fimg, out["image"]))
            leg["samples_equal_to_headline_schedule"] = bool(int(fst[5]) == samples_per_frame)
            if args.precision == "f32":
                leg["head_frac"] = round(FLOP_PER_SAMPLE * int(fst[5]) * args.steps / (fms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)
                leg["head_issued_frac"] = round(ISSUED_FLOP_PER_ROW * int(fst[72]) * args.steps / (fms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)
            result["reference_schedule"] = leg
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["reference_schedule"] = repr(exc)
    if world == 1 and args.precision == "f32" and not args.no_fp16_leg:
        try:
            # the reference's opt.fp16 arithmetic (torch autocast) on the f16 matrix cores: a different rounding sequence, so it is
            # reported beside the bit-exact f32 headline, with its distance from the f32 image
            h16 = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, bound=1.0, device=device, precision="f16")
            leg, img16, _, _ = side_leg(h16, args.budget_factor, args.n_step_cap)
            leg8, img16b, _, _ = side_leg(h16, *REF_SCHEDULE)
            diff = (img16 - out["image"]).double()
            mse16 = float((diff ** 2).mean())
            leg.update(dtype="f16 (f32 accumulate, torch-autocast rounding)", kernel="lz_k_triplane_head_f16",
                       max_abs_diff_vs_f32_image=float(diff.abs().max()), psnr_vs_f32_image_db=round(-10 * np.log10(max(mse16, 1e-300)), 2),
                       reference_schedule_value=leg8["value"], reference_schedule_ms_per_step=leg8["ms_per_step"],
                       reference_schedule_image_equal=bool(torch.equal(img16, img16b)))
            result["fp16_head"] = leg
            del h16
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["fp16_head"] = repr(exc)
    if world == 1 and not args.no_dense192:
        try:
            # SURVEY 8d "dense-192 micro-benchmark": the NOMINAL 512 x 512 x 192 = 50.33 M samples (uniform points in [-1,1]^3, the ray
            # directions, delta = 2 sqrt(3) / 192) straight through encode -> MLP (fused head) -> composite_rays_train_triplane
            from lzzx_nerf_amd import raymarching as R
            S = args.max_steps
            gd = torch.Generator(device=device).manual_seed(5)
            M = N * S
            xyz = torch.rand(M, 3, device=device, generator=gd) * 2 - 1
            dirs_d = rays_d.repeat_interleave(S, dim=0)
            dt = float(2 * np.sqrt(3) / S)
            deltas = torch.empty(M, 2, device=device)
            deltas[:, 0] = dt
            deltas[:, 1] = (torch.arange(M, device=device) % S).float() * dt + 2.35
            rays_tbl = torch.stack([torch.arange(N, device=device), torch.arange(N, device=device) * S, torch.full((N,), S, device=device)],
                                   1).int().contiguous()
            outd = tuple(torch.empty(s, device=device) for s in ((M,), (M, 3), (M, 1), (M, 1), (M, 1)))

            def dense():
                sg, rg, aa, ae, un = head.forward(xyz, dirs_d, enc_a, ind, eye, testing=True, out=outd)
                return R.composite_rays_train_triplane(sg, rg, aa.view(-1), ae.view(-1), un.view(-1), deltas, rays_tbl)

            for _ in range(2):
                dense()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                comp = dense()
            torch.cuda.synchronize()
            dms = (time.perf_counter() - t0) / 5 * 1e3
            result["dense192"] = dict(samples=M, ms=round(dms, 3), samples_per_s=round(M / dms * 1e3, 1), rays_per_s=round(N / dms * 1e3, 1),
                                      note="nominal 512x512x192 samples: uniform points -> fused head -> composite_rays_train_triplane forward",
                                      image_mean=float(comp[5].mean()))
            del xyz, dirs_d, deltas, rays_tbl, outd, comp
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["dense192"] = repr(exc)
    if not args.no_grid_roofline and world == 1:
        try:
            result["roofline_gridencoder"] = grid_roofline(device)
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["roofline_gridencoder"] = repr(exc)
    if not args.no_train and world == 1:
        try:
            result["train_step"] = train_bench(args, device, P, golden, bits)
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["train_step"] = repr(exc)
    if train_dp is not None:
        result["train_step"] = train_dp
    if world == 1 and not args.no_occupancy and args.precision == "f32":
        try:
            # the occupancy-grid maintenance of update_extra_state (renderer.py:699-766) as 5 launches, no sync
            from lzzx_nerf_amd.occupancy import update_density_grid
            dg = torch.zeros(1, 128 ** 3, device=device)
            bf = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=device)
            nz = torch.rand(1, 128 ** 3, 3, device=device, generator=torch.Generator(device=device).manual_seed(2))
            for _ in range(2):
                update_density_grid(head, dg, bf, enc_a, eye, bound=1.0, noise=nz)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                mean_d, _ = update_density_grid(head, dg, bf, enc_a, eye, bound=1.0, noise=nz)
            torch.cuda.synchronize()
            result["occupancy_grid_update"] = dict(ms=round((time.perf_counter() - t0) / 5 * 1e3, 3), cells=128 ** 3, cascade=1,
                                                   mean_density=float(mean_d), launches=5, host_syncs=0)
            del dg, bf, nz
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["occupancy_grid_update"] = repr(exc)
    if world == 1 and not args.no_occupancy:
        try:
            # torso branch of the frame (run_torso + forward_torso) as one kernel, 512 x 512 pixels, random weights
            from lzzx_nerf_amd.torso import FusedTorso
            from lzzx_nerf_amd.gridencoder import grid_offsets
            rngt = np.random.default_rng(7)
            offs = np.asarray(grid_offsets(2, 16, np.exp2(np.log2(2048 / 16) / 15), 16, 16))
            lin = lambda n, k: torch.from_numpy((rngt.uniform(-1, 1, (n, k)) / np.sqrt(k)).astype(np.float32))
            sdt = {"anchor_points": torch.tensor([[0.01, 0.01, 0.1, 1], [-0.1, -0.1, 0.1, 1], [0.1, -0.1, 0.1, 1]]),
                   "torso_deform_net.net.0.weight": lin(32, 84), "torso_deform_net.net.1.weight": lin(32, 32),
                   "torso_deform_net.net.2.weight": lin(2, 32), "torso_net.net.0.weight": lin(32, 116), "torso_net.net.1.weight": lin(32, 32),
                   "torso_net.net.2.weight": lin(4, 32), "torso_encoder.offsets": torch.from_numpy(offs.astype(np.int32)),
                   "torso_encoder.embeddings": torch.from_numpy(rngt.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32))}
            torso = FusedTorso(sdt, device=device)
            lin1 = torch.linspace(-1, 1, H, device=device)
            bgc = torch.stack(torch.meshgrid(lin1, lin1, indexing="xy"), -1).reshape(-1, 2).contiguous()
            enc_anchor = torso.encode_anchor(dev(pose)[None])
            indt = torch.zeros(8, device=device)
            for _ in range(3):
                torso(bgc, ind_code=indt, enc_anchor=enc_anchor)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ta, tc, _ = torso(bgc, ind_code=indt, enc_anchor=enc_anchor)
            torch.cuda.synchronize()
            result["torso_branch"] = dict(ms=round((time.perf_counter() - t0) / 20 * 1e3, 4), pixels=H * W, launches=1,
                                          note="all pixels queried (no 2-D occupancy mask); 5.4 kMAC per pixel on the VALU")
            del torso, bgc
            # encode_audio (AudioNet on 8 HuBERT windows [8, 1024, 16] + AudioAttNet) as one launch, random weights
            from lzzx_nerf_amd.audio import FusedAudioEncoder
            ga = torch.Generator().manual_seed(3)
            sda = {}
            for idx, (ci, co) in zip((0, 2, 4, 6), ((1024, 32), (32, 32), (32, 64), (64, 64))):
                sda[f"audio_net.encoder_conv.{idx}.weight"] = (torch.rand(co, ci, 3, generator=ga) * 2 - 1) / (3 * ci) ** 0.5
                sda[f"audio_net.encoder_conv.{idx}.bias"] = torch.zeros(co)
            for idx, (ci, co) in zip((0, 2), ((64, 64), (64, 32))):
                sda[f"audio_net.encoder_fc1.{idx}.weight"] = (torch.rand(co, ci, generator=ga) * 2 - 1) / ci ** 0.5
                sda[f"audio_net.encoder_fc1.{idx}.bias"] = torch.zeros(co)
            for idx, (ci, co) in zip((0, 2, 4, 6, 8), ((32, 16), (16, 8), (8, 4), (4, 2), (2, 1))):
                sda[f"audio_att_net.attentionConvNet.{idx}.weight"] = (torch.rand(co, ci, 3, generator=ga) * 2 - 1) / (3 * ci) ** 0.5
                sda[f"audio_att_net.attentionConvNet.{idx}.bias"] = torch.zeros(co)
            sda["audio_att_net.attentionNet.0.weight"] = torch.eye(8)
            sda["audio_att_net.attentionNet.0.bias"] = torch.zeros(8)
            aenc = FusedAudioEncoder(sda, device=device)
            auds = torch.randn(8, 1024, 16, device=device, generator=torch.Generator(device=device).manual_seed(4))
            for _ in range(3):
                aenc(auds)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                aenc(auds)
            torch.cuda.synchronize()
            result["audio_frontend"] = dict(ms=round((time.perf_counter() - t0) / 20 * 1e3, 4), windows=8, dim_in=1024, launches=2)
            del aenc, auds
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["torso_audio"] = repr(exc)
    # ---- CPU baseline: the checker arranged like the reference loop, on a bounded sub-frame of the SAME rays ----
    if not args.no_cpu_baseline and world == 1:   # rank 0, N = 1 only
        try:
            from oracle.head import TriplaneSpec
            from oracle.render import render_inference
            stride = max(1, H // 64)
            sel = (np.arange(0, H, stride)[:, None] * W + np.arange(0, W, stride)[None, :]).reshape(-1)
            ro_c, rd_c = rays_o.cpu().numpy()[sel], rays_d.cpu().numpy()[sel]
            st = {}
            render_inference(TriplaneSpec(1.0), P, ro_c[:256], rd_c[:256], bits, golden["net_enc_a"], golden["net_ind"], golden["net_eye"],
                             max_steps=args.max_steps)  # warm-up (page in, OpenMP team)
            tc = time.perf_counter()
            ref = render_inference(TriplaneSpec(1.0), P, ro_c, rd_c, bits, golden["net_enc_a"], golden["net_ind"], golden["net_eye"],
                                   max_steps=args.max_steps, stats=st, budget_factor=args.budget_factor, n_step_cap=args.n_step_cap)
            tc = time.perf_counter() - tc
            cpu_samples = int(st["samples_per_ray"].sum())
            gpu_img = out["image"].cpu().numpy()[sel]
            mse = float(((gpu_img.astype(np.float64) - ref["image"]) ** 2).mean())
            psnr = float("inf") if mse == 0 else -10 * np.log10(mse)
            result["cpu_baseline"] = dict(value=round(cpu_samples / tc, 1), unit="samples/s", cores=len(os.sched_getaffinity(0)), kind="port",
                                          sample=f"{len(sel)} rays (every {stride}th pixel of the same frame), {cpu_samples} samples, "
                                                 f"{tc:.1f} s; checker arranged like run_cuda_for_inference (renderer.py:495-548), OpenMP")
            result["psnr_vs_checker_db"] = psnr if np.isfinite(psnr) else "inf"
            result["max_abs_diff_vs_checker"] = float(np.abs(gpu_img - ref["image"]).max())
            cnt = renderer.render(rays_o, rays_d, enc_a, ind, eye, dt_gamma=1 / 256, max_steps=args.max_steps, T_thresh=1e-4,
                                  count_samples=True)["ray_counts"].cpu().numpy()[sel]
            result["sample_counts_equal"] = bool(np.array_equal(cnt.astype(np.int64), st["samples_per_ray"]))
        except Exception as exc:   # reported, never fatal for the line
            result.setdefault("leg_errors", {})["cpu_baseline"] = repr(exc)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
