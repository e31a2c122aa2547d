"""Everything bench.py reports BESIDE its contract line: the side legs (other schedules / precisions / configs), the cfg3 training
step, the grid-encoder roofline, the CPU baseline, and the helpers that time a frame job.  bench.py imports this module, builds the full
result with it, writes that to bench_detail.json and prints only tools/bench_contract.compact(result) on stdout.

Only `cpu_baseline` may import the checker under oracle/ (tests/test_cabi.py enforces it)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 46368          # 23 184 MAC per sample, inference head
ISSUED_FLOP_PER_ROW = 361 * 2048 // 16   # the head issues 361 v_mfma_f32_16x16x4_f32 (2048 FLOP each) per 16-row slice
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, v_mfma_f32_16x16x4_f32
HBM_PEAK_GBS = 8000.0
GRID_ORDERED_PMC_SUMMARY = "r4_grid_ordered_pmc_summary.json"   # tools/profile_grid_ordered.sh
GRID_REQ_SUMMARY = "r5_grid_request_rate.json"   # tools/profile_grid_req.sh: L1 -> L2 requests / lane accesses per launch of the ordered cfg2 gather legs + the probe's peaks
GRID_PMC_SUMMARY = "r3_grid_pmc_summary.json"   # tools/profile_grid.sh over the CURRENT kernels (a summary of an older round describes code that no longer exists)


# kernels behind each grid_roofline case and the batch size tools/grid_bench.py profiled them at (tools/profile_grid.sh)
_GRID_PMC = {"triplane_plane_D2_L12_C1_f32": (["lz_k_grid_forward_lds<float, 2u, 1u>"], 1 << 22),
             "hashgrid_D3_L16_C2_f32": (["lz_k_grid_forward_lmp<float, 3u, 2u>", "lz_k_grid_untile"], 1 << 23),
             "hashgrid_D3_L16_C2_f16": (["lz_k_grid_forward_lmp<__half, 3u, 2u>", "lz_k_grid_untile"], 1 << 23),
             "triplane_plane_D2_L12_C1_f32_backward": (["lz_k_grid_backward_lds_fx<2u, 1u>"], 1 << 22)}


def _grid_traffic(tag, B):
    """HBM-side bytes per launch at batch size B from the committed PMC summary (KiB per launch at the profiled batch size, scaled
    per sample; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 -- calibrated there for wide streaming reads, so an
    upper bound for the gather-dominated kernels).  None when the summary is absent."""
    for suffix, order in (("_ray_ordered", "ray"), ("_march_order", "march")):
        if tag.endswith(suffix):   # the ordered legs share kernel names with the random one: profiled one process per case (tools/profile_grid_ordered.sh)
            try:
                case = json.load(open(os.path.join(ROOT, "profiles", GRID_ORDERED_PMC_SUMMARY)))[order + ("_f16" if "f16" in tag else "_f32")]
                kib = sum(2 * case["FETCH_SIZE"][k]["avg_per_launch"] + case["WRITE_SIZE"][k]["avg_per_launch"] for k in case["FETCH_SIZE"] if "lz_k_grid" in k)
                return round(kib * 1024 / (1 << 23) * B)
            except (OSError, KeyError, ValueError):
                return None
    path = os.path.join(ROOT, "profiles", GRID_PMC_SUMMARY)
    if not os.path.exists(path) or tag not in _GRID_PMC:
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


def _grid_request_rate(tag, ms):
    """the cfg2 gather against REQUEST-rate yardsticks (with half tables the byte roofline is the wrong one -- the kernel
    makes the f32 kernel's line fills and lane accesses for half the bytes): lane loads per second against the chip's L1-hit scatter peak and
    line fills per second against its L2-resident fill peak, both measured by tools/probes/l1_fill_probe.hip; counters per launch from the
    committed PMC pass of the same case (one process per case, tools/profile_grid_req.sh).  None when the summary is absent."""
    case = ("march" if tag.endswith("_march_order") else "ray" if tag.endswith("_ray_ordered") else None)
    if case is None:
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", GRID_REQ_SUMMARY)))
        c, pk = d["cases"][case + ("_f16" if "f16" in tag else "_f32")], d["peaks"]
        acc, fills = c["TCP_TOTAL_ACCESSES_per_launch"] / (ms * 1e-3), c["TCP_TCC_READ_REQ_per_launch"] / (ms * 1e-3)
        return dict(bound="l1 lane-address rate", achieved=round(acc, -7), peak=pk["l1_hit_scatter_lane_loads_per_s"], unit="lane loads/s",
                    frac=round(acc / pk["l1_hit_scatter_lane_loads_per_s"], 4), line_fills_per_s=round(fills, -7),
                    frac_of_l2_resident_fill_peak=round(fills / pk["l2_resident_line_fills_per_s"], 4),
                    lane_loads_per_launch=c["TCP_TOTAL_ACCESSES_per_launch"], line_fills_per_launch=c["TCP_TCC_READ_REQ_per_launch"],
                    source="profiles/" + GRID_REQ_SUMMARY)
    except (OSError, KeyError, ValueError, ZeroDivisionError):
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
            ("hashgrid_D3_L16_C2_f32_march_order", dict(desired_resolution=2048), 12 + 16 * 8 * 2 * 4 + 128, 1 << 23, "fwd_march"),
            ("hashgrid_D3_L16_C2_f16", dict(desired_resolution=2048), 12 + 16 * 8 * 2 * 2 + 64, 1 << 23, "fwd16"),
            ("hashgrid_D3_L16_C2_f16_ray_ordered", dict(desired_resolution=2048), 12 + 16 * 8 * 2 * 2 + 64, 1 << 23, "fwd_rays16"),
            ("hashgrid_D3_L16_C2_f16_march_order", dict(desired_resolution=2048), 12 + 16 * 8 * 2 * 2 + 64, 1 << 23, "fwd_march16"),
            ("triplane_plane_D2_L12_C1_f32_backward", tri, 8 + 48 + 12 * 4 * 4 * 2, 1 << 22, "bwd")):
        enc = GridEncoder(**kw).to(device)
        enc.embeddings.data.uniform_(-1, 1, generator=g)
        half = mode.endswith("16")
        mode = mode[:-2] if half and mode != "fwd16" else mode
        if mode == "fwd_march":   # BASELINE cfg2 as the INFERENCE loop hands it to the encoder (renderer.py:513-521): per iteration every
            # alive ray contributes n_step = 8 consecutive samples, rays in pixel order -> [iteration][ray][8 steps]; 16 iterations of the
            # 256 x 256 frame = 2^23 samples (all rays kept alive: the densest case)
            from lzzx_nerf_amd.synthetic import synthetic_camera
            from lzzx_nerf_amd.utils import frame_rays
            pose, intr = synthetic_camera(256, 256)
            ro, rd = frame_rays(torch.from_numpy(pose).to(device), intr, 256, 256)
            t = torch.linspace(2.35, 4.35, 128, device=device).view(16, 1, 8)                       # [iteration, 1, step]
            p = ro[None, :, None, :] + rd[None, :, None, :] * t[..., None]                        # [16, rays, 8, 3]
            x = ((p.clamp(-1, 1) + 1) / 2).reshape(-1, 3).contiguous()
        elif mode == "fwd_rays":   # march_rays_train order: 256 x 256 rays x 128 consecutive samples, ray-major
            from lzzx_nerf_amd.synthetic import synthetic_camera
            from lzzx_nerf_amd.utils import frame_rays
            pose, intr = synthetic_camera(256, 256)
            ro, rd = frame_rays(torch.from_numpy(pose).to(device), intr, 256, 256)
            t = torch.linspace(2.35, 4.35, 128, device=device)
            x = (((ro[:, None, :] + rd[:, None, :] * t[None, :, None]).clamp(-1, 1) + 1) / 2).reshape(-1, 3).contiguous()
        else:
            x = torch.rand(B, enc.input_dim, device=device, generator=g)
        emb = enc.embeddings.data.half() if half else enc.embeddings.data
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
        rr = _grid_request_rate(tag, ms)
        if rr is not None:
            res[tag]["request_rate"] = rr
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


def train_dtype(args):
    if args.train_mlp != "fused" or "f16" not in (args.train_records, args.train_forward, args.train_backward):
        return "f32"
    fwd = "f16 forward (autocast arithmetic)" if args.train_forward == "f16" else "f32 forward"
    bwd = "f16 data gradient (autocast arithmetic)" if args.train_backward == "f16" else "f32 data gradient"
    return f"{fwd}, {bwd}, f16 weight-gradient operands, f32 accumulation; GradScaler(65536)"


def train_bench(args, device, P, golden, bits, rank=0, world=1, n_rays=None):
    """BASELINE cfg3: one training step (fwd + bwd + Adam) on `--train-rays` random rays of the 512x512 frame through the
    operator API as the reference's run_cuda arranges it (renderer.py:279-304).  The MLP GEMMs are torch/rocBLAS here;
    everything else is this repo's HIP kernels.  Reported beside the headline line, never instead of it."""
    from lzzx_nerf_amd import dist as D
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd.synthetic import synthetic_camera
    from lzzx_nerf_amd.utils import frame_rays
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    H = W = args.size
    pose, intr = synthetic_camera(H, W)
    ro, rd = frame_rays(dev(pose), intr, H, W)
    g = torch.Generator(device=device).manual_seed(0)
    n_rays = min(n_rays or args.train_rays, H * W)
    sel = torch.randperm(H * W, device=device, generator=g)[:n_rays]
    target = torch.rand(n_rays, 3, device=device, generator=g)
    lo, hi = D.shard_bounds(n_rays, rank, world)   # data parallel over the sampled rays (world == 1: everything)
    ro, rd, target = ro[sel[lo:hi]].contiguous(), rd[sel[lo:hi]].contiguous(), target[lo:hi].contiguous()
    if args.train_mlp == "fused":   # one kernel forward, one kernel for the backward data chain (lzzx_nerf_amd/head_train.py)
        from lzzx_nerf_amd.head_train import FusedTriplaneTrainHead
        all16 = args.train_forward == "f16" and args.train_backward == "f16"
        net = FusedTriplaneTrainHead(P, bound=1.0, record=not args.train_recompute, record_dtype=args.train_records, forward_dtype=args.train_forward,
                                     backward_dtype=args.train_backward, recompute_mlp=(not getattr(args, "train_keep_records", False)) if all16 else None).to(device)
    else:
        net = TriplaneTrainNet(P, device, mlp=args.train_mlp)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, eps=1e-15, fused=True)
    bucket = D.GradientBucket(net.parameters()) if world > 1 else None   # every .grad a view of one flat buffer: ONE all-reduce per step
    enc_a, ind, eye = dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"])
    aabb = dev(np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32))
    bitfield = dev(bits)
    ctr = torch.zeros(2, dtype=torch.int32, device=device)
    n_samples = [0]
    # The reference sizes the sample buffers from a running average of the step counter once it has one (mean_count > 0,
    # renderer.py:814-818 -> raymarching.py:221-228): no D2H copy in the step.  Only while mean_count <= 0 (the first steps) or with
    # force_all_rays does its wrapper trim by counter[0].item().  The first warm-up step below is that start-up case and yields the
    # count; every later step, the timed ones included, runs the steady state (noise perturbs t0 by less than one step, so the
    # count of a step stays within the 128-row padding of the first one's plus a margin).
    mean_count = [-1]
    half = args.train_mlp == "fused" and "f16" in (args.train_records, args.train_forward, args.train_backward)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0) if (half and world == 1) else None
    layout = getattr(args, "train_layout", "step")

    def step():
        nears, fars = R.near_far_from_aabb(ro, rd, aabb, 0.05)
        ctr.zero_()
        xyzs, dirs, deltas, rays = R.march_rays_train(ro, rd, 1.0, bitfield, 1, 128, nears, fars, ctr, mean_count[0], True, 128,
                                                      mean_count[0] <= 0, 1 / 256, args.max_steps, layout=layout)
        if mean_count[0] <= 0:
            mean_count[0] = int(xyzs.shape[0]) + n_rays // 64   # margin: a perturbed ray gains or loses at most one sample
        sigma, rgb, a0, a1, unc = net(xyzs, dirs, enc_a, ind, eye)
        if args.train_mlp == "fused":
            a0, a1, unc = a0.squeeze(-1), a1.squeeze(-1), unc.squeeze(-1)      # views: their backward is a view too (a select's is zeros + copy)
        ws, a0s, a1s, us, dep, img = R.composite_rays_train_triplane(sigma, rgb, a0, a1, unc, deltas, rays)
        loss = ((img + (1 - ws).unsqueeze(-1) - target) ** 2).mean() + 1e-4 * a0s.mean() + 1e-4 * a1s.mean() + 1e-3 * us.mean()
        if scaler is not None:   # half operands: loss scaling exactly as the reference's trainer does it (TrainerUtil.py:103, 865-870)
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            return loss
        if bucket is None:
            opt.zero_grad(set_to_none=True)
            loss.backward()
        else:
            bucket.zero()
            loss.backward()
            bucket.all_reduce()
        opt.step()
        return loss

    k_steps = args.steps if n_rays <= 65536 else max(5, args.steps // 2)
    for _ in range(max(args.warmup, 2)):
        step()
    torch.cuda.synchronize()
    overflow = int(ctr[0].item()) > mean_count[0]     # rays dropped because the buffer was too small (raymarching.cu:457)? must be False
    if world > 1:
        torch.distributed.barrier()
    ms0 = torch.cuda.memory_stats(device)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(k_steps + 1)]   # step boundaries on the stream (no sync inside the loop)
    import gc
    gc.collect()
    gc.disable()       # as timeit does: a generation-2 collection in the middle of a step is a host stall of tens of milliseconds
    for _ in range(4):      # the synchronize / .item() / gc.collect() above let the GPU idle and its clock drop: the first steps behind them
        step()              # ran 5.35, 4.57, 4.47, 4.39 ms against 4.25 steady (LZ_TRAIN_STEP_TRACE=1); these four are not timed
    torch.cuda.synchronize()   # (a bare wait: the queue is full again microseconds later)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(k_steps):
        loss = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k_steps
    gc.enable()
    per_step_order = [marks[i].elapsed_time(marks[i + 1]) for i in range(k_steps)]
    if os.environ.get("LZ_TRAIN_STEP_TRACE"):
        log("train steps ms: " + " ".join(f"{t:.2f}" for t in per_step_order))
    per_step = sorted(per_step_order)
    ms1 = torch.cuda.memory_stats(device)
    # hipMalloc / hipFree calls of torch's caching allocator inside the timed region (each is a device-wide stall of milliseconds at these
    # buffer sizes): must be 0 for the number to mean anything
    dev_allocs = int(ms1.get("segment.all.allocated", 0) - ms0.get("segment.all.allocated", 0))
    dev_frees = int(ms1.get("segment.all.freed", 0) - ms0.get("segment.all.freed", 0))
    n_samples[0] = int(ctr[0].item())      # samples of the last step (after the timed region)
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
                         "scatter-add) -> Adam; sample buffers sized by mean_count like the reference's steady state (no D2H copy in the step)",
                sample_rows={"step": "step-major groups of 64 rays in locality order (lz_ray_sort_keys -> sort -> lz_march_rays_train_grouped, all inside the timed step)",
                             "ray": "ray-major (the reference's)"}[layout],
                rays=n_rays, samples_per_step=int(n_samples[0]), steps=k_steps, sample_buffer_overflow=overflow,
                device_allocations_in_timed_region=dev_allocs, device_frees_in_timed_region=dev_frees,
                ms_per_step_median=round(per_step[len(per_step) // 2], 3), ms_per_step_min=round(per_step[0], 3), ms_per_step_max=round(per_step[-1], 3),
                ms_per_step=round(dt * 1e3, 3), samples_per_s=round(n_samples[0] / dt, 1), rays_per_s=round(n_rays / dt, 1),
                loss=float(loss.detach()), dtype=train_dtype(args), mlp={"fused": "fused head forward + backward kernels, " + ("activations recomputed in the backward (csrc/lz_head.hip, lz_head_bwd.hip)" if args.train_recompute else ("the forward keeps only the enc_x halves (80 B per sample), the backward kernel recomputes the MLP from them (csrc/lz_head_fwd16_chain.h, lz_head_rec.hip RC)" if (args.train_forward == "f16" and args.train_backward == "f16" and not getattr(args, "train_keep_records", False)) else "forward records, backward starts from the record (csrc/lz_head_rec.hip" + (", lz_head_rec16.hip" if args.train_forward == "f16" else "") + "), " + ("f16" if "f16" in (args.train_forward, args.train_backward) else args.train_records) + " records")), "lz": "csrc/lz_linear.hip (MFMA f32)",
                     "torch": "torch/rocBLAS"}[args.train_mlp])


def ray_order_perm(kind, H, W, device):
    """pixel permutations for the ray-order experiment: 'tile<w>x<h>' (row-major tiles, row-major inside), 'morton', 'random'"""
    y, x = np.divmod(np.arange(H * W), W)
    if kind.startswith("tile"):
        tw, th = (int(v) for v in kind[4:].split("x"))
        key = ((y // th) * (W // tw) + x // tw) * (tw * th) + (y % th) * tw + x % tw
    elif kind == "morton":
        def spread(v):
            v = v.astype(np.uint64)
            out = np.zeros_like(v)
            for b in range(16):
                out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b)
            return out
        key = spread(x) | (spread(y) << np.uint64(1))
    elif kind == "random":
        key = np.random.default_rng(0).permutation(H * W)
    else:
        raise ValueError(kind)
    return torch.from_numpy(np.argsort(key, kind="stable")).to(device)


class FrameJob:
    """one rank's share of the per-step work: ray generation for its pixels -> render -> (tile all-gather)"""

    def __init__(self, renderer, H, W, pose, intr, cond, max_steps, rank, world, shard, tiles, gather, device, shard_of=0, via="collective"):
        from lzzx_nerf_amd import dist as D
        self.r, self.H, self.W, self.intr, self.cond, self.max_steps, self.gather_fmt = renderer, H, W, intr, cond, max_steps, gather
        self.pose = torch.from_numpy(np.ascontiguousarray(pose)).to(device)
        self.shard = shard if world > 1 else "frame"
        if shard_of > 1:          # single-GPU rehearsal of one rank of an N-way sharded frame
            self.sf = D.ShardedFrame(H, W, 0, shard_of, tiles, device)
            self.sf.gatherer = None
        elif world > 1 and shard == "frame":
            self.sf = D.ShardedFrame(H, W, rank, world, tiles, device, dtype=torch.uint8 if gather == "rgb24" else torch.float32, via=via,
                                     cap=getattr(renderer, "cap", "reference"))
            self.sf.configure(renderer)      # cap "reference": the frame's ray count + the histogram all-reduce between the two phases
        else:
            self.sf = D.ShardedFrame(H, W, 0, 1, device=device)
            if world > 1:         # clip mode: every rank contributes a whole frame to the gathered batch
                self.sf.gatherer = D.TileGatherer(H * W, 3, torch.uint8 if gather == "rgb24" else torch.float32, device)
        self.n_rays = self.sf.n_local

    def step(self):
        rays_o, rays_d = self.sf.rays(self.pose, self.intr)          # ray generation is part of the step (north_star lists it on the path)
        if os.environ.get("LZ_EXP_RAY_ORDER"):                       # experiment: the same rays handed over in another order (tools/exp_ray_order.sh)
            if getattr(self, "_perm", None) is None:
                self._perm = ray_order_perm(os.environ["LZ_EXP_RAY_ORDER"], self.H, self.W, rays_o.device)
            rays_o, rays_d = rays_o[self._perm].contiguous(), rays_d[self._perm].contiguous()
        enc_a, ind, eye = self.cond
        out = self.r.render(rays_o, rays_d, enc_a, ind, eye, dt_gamma=1 / 256, max_steps=self.max_steps, T_thresh=1e-4,
                            rgb24=self.gather_fmt == "rgb24")
        tile = out["image_rgb24"] if self.gather_fmt == "rgb24" else out["image"]
        return out, self.sf.gather(tile)


def timed(job, steps, warmup, world, device, timing=True):
    """W warm-up steps, then exactly K steps bracketed by barrier + synchronize; max over ranks.  -> (dt seconds, head launch ms list, last out, last tiles)"""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    import gc
    gc.collect()       # before the warm-up steps: a collection between them and the timed region is milliseconds of idle GPU (clock ramp)
    gc.disable()       # as timeit does
    for _ in range(warmup):
        job.step()
    barrier()
    if timing:
        job.r.timing_start(steps * job.max_steps + 16)   # HIP event pair around every head launch, on the launch stream
    t0 = time.perf_counter()
    for _ in range(steps):
        out, tiles = job.step()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    head_ms = job.r.timing_stop() if timing else []
    job.dt_own = dt            # this rank's own clock (the return value is the max over the ranks)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, head_ms, out, tiles


def frame_stats(out, world, device):
    """(samples, iterations, head rows) of the last frame, summed (max for iterations) over ranks"""
    st = out["state"].cpu().numpy()
    v = torch.tensor([float(st[5]), float(st[72])], dtype=torch.float64, device=device)
    it = torch.tensor([float(st[6])], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(v)
        torch.distributed.all_reduce(it, op=torch.distributed.ReduceOp.MAX)
    return int(v[0].item()), int(it.item()), int(v[1].item())


def cpu_baseline(args, P, golden, bits_np, gpu_image_of, renderer_counts_of):
    """SURVEY 8(d) / BASELINE.md 3: the reference has no CPU renderer, so the baseline is the CPU checker's kernels (C, OpenMP) arranged
    exactly like run_cuda_for_inference (renderer.py:495-548) with the reference's pure-torch MLP arrangement on CPU tensors
    (oracle.head.head_forward_torch: F.linear stacks, fp32, torch intra-op threads = all cores).  cfg1 = whole 64x64 / 32-step frames,
    median of 5 after a warm-up; cfg3 = the WHOLE 512x512 / 192-step frame (262 144 rays, 23.9 M samples, reference schedule), median of 5
    after one warm-up (about 70 s on the 16 cores of a GPU box) -- `--cpu-baseline-sample` times every 4th pixel in both
    directions instead (16 384 rays, median of 3: the bounded sample of rounds 1-2).  Also yields PSNR / sample-count parity of the GPU
    image against the bit-pinned checker on every 8th pixel."""
    from oracle import oracle as O
    from oracle.head import TriplaneSpec, head_forward_torch
    from oracle.render import render_inference
    from lzzx_nerf_amd.synthetic import ones_bitfield, synthetic_camera
    cores = O.host_cores()          # affinity mask capped by the cgroup CPU quota: teams larger than the quota only fight each other
    torch.set_num_threads(cores)
    O.set_threads(cores)
    spec = TriplaneSpec(1.0)
    cond = (golden["net_enc_a"], golden["net_ind"], golden["net_eye"])
    res = {}

    def run(H, W, stride, max_steps, reps, bits):
        pose, intr = synthetic_camera(H, W)
        sel = (np.arange(0, H, stride)[:, None] * W + np.arange(0, W, stride)[None, :]).reshape(-1)
        r = O.get_rays_batched(pose[None], intr, H, W, sel)
        ro, rd = r["rays_o"][0], r["rays_d"][0]
        st, times = {}, []
        f = lambda: render_inference(spec, P, ro, rd, bits, *cond, max_steps=max_steps, stats=st, head=head_forward_torch)
        f()                                             # warm-up (page in, OpenMP / torch thread teams)
        for _ in range(reps):
            t = time.perf_counter()
            f()
            times.append(time.perf_counter() - t)
        med = float(np.median(times))
        ns = int(st["samples_per_ray"].sum())
        return dict(rays=len(sel), samples=ns, s_per_frame=round(med, 4), samples_per_s=round(ns / med, 1), rays_per_s=round(len(sel) / med, 1),
                    runs=reps), sel
    log(f"cpu baseline: cfg1 on {cores} cores")
    cfg1, _ = run(64, 64, 1, 32, 5, ones_bitfield())
    log(f"cpu baseline: cfg1 {cfg1['s_per_frame']} s/frame; cfg3 sample")
    res["cfg1_64x64x32"] = cfg1
    H = W = args.size
    stride = max(1, H // 128) if args.cpu_baseline_sample else 1
    reps = 3 if args.cpu_baseline_sample else 5
    cfg3, sel = run(H, W, stride, args.max_steps, reps, bits_np)
    res["cfg3_sample" if args.cpu_baseline_sample else "cfg3_whole_frame"] = cfg3
    log(f"cpu baseline: cfg3 {cfg3['s_per_frame']} s/frame ({cfg3['rays']} rays)")
    what = f"on every {stride}th pixel in both directions = " if stride > 1 else "whole = "
    base = dict(value=cfg3["samples_per_s"], unit="samples/s", cores=cores, kind="port",
                sample=f"cfg3 frame ({H}x{W}, max_steps {args.max_steps}) {what}{cfg3['rays']} rays, "
                       f"{cfg3['samples']} samples, median of {reps} after a warm-up: {cfg3['s_per_frame']} s; checker kernels (C, OpenMP) arranged like "
                       "run_cuda_for_inference (renderer.py:495-548) + the reference's torch-CPU MLP arrangement (network.py:73-94), fp32, "
                       f"torch.set_num_threads({cores})",
                legs=res, cfg1_samples_per_s=cfg1["samples_per_s"], cfg1_s_per_frame=cfg1["s_per_frame"])
    # parity of the GPU frame against the BIT-PINNED checker (order-fixed MLP) on every 8th pixel
    stride8 = max(1, H // 64)
    sel8 = (np.arange(0, H, stride8)[:, None] * W + np.arange(0, W, stride8)[None, :]).reshape(-1)
    pose, intr = synthetic_camera(H, W)
    r = O.get_rays_batched(pose[None], intr, H, W, sel8)
    st = {}
    ref = render_inference(spec, P, r["rays_o"][0], r["rays_d"][0], bits_np, *cond, max_steps=args.max_steps, stats=st,
                           budget_factor=args.budget_factor, n_step_cap=args.n_step_cap)
    gpu_img = gpu_image_of(sel8)
    mse = float(((gpu_img.astype(np.float64) - ref["image"]) ** 2).mean())
    parity = dict(psnr_vs_checker_db="inf" if mse == 0 else round(-10 * np.log10(mse), 2),
                  max_abs_diff_vs_checker=float(np.abs(gpu_img - ref["image"]).max()),
                  sample_counts_equal=bool(np.array_equal(renderer_counts_of(sel8).astype(np.int64), st["samples_per_ray"])),
                  parity_sample=f"{len(sel8)} rays (every {stride8}th pixel), {int(st['samples_per_ray'].sum())} samples")
    return base, parity


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries only the JSON line)"""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


REF_SCHEDULE = (1, 8)   # renderer.py:513
PMC_SUMMARY = "r5_final_pmc_summary.json"   # written by tools/profile_bench.sh from rocprofv3 --pmc passes of this same command
F16_SLICE_MFMAS = 30    # v_mfma_f32_32x32x16_f16 per 16 sample rows (60 per 32-sample slice of lz_head16w_slice; rounds 2-4: 59 16x16x32 per 16)
F16_MFMA_CYCLES = 32    # matrix-pipe cycles of one v_mfma_f32_32x32x16_f16 (MI355X_MICROARCH.md)


F16_PMC_SUMMARY = "r5_f16_head_pmc_summary.json"   # tools/profile_bench.sh f16 over the current kernel
F16_MFMA_PEAK_TFLOPS = 2500.0                      # dense f16 (MI355X_MICROARCH.md; AMD's 5 PF figure includes 2:1 sparsity)


def _pmc_per_launch(rec):
    """counter value per launch from a tools/summarize_pmc.py record: the mean over all launches but the largest one (the profiled command's
    first, cold launch now and then reads several times the others' WRITE_SIZE), the plain mean when there are fewer than three"""
    n = rec["launches"]
    return (rec["sum"] - rec["max"]) / (n - 1) if n >= 3 else rec["avg_per_launch"]


def f16_head_roofline(samples, rows, steps, head_total_ms, n_launch, launches_with_work, dt, fused=True, n_rays=None, packed_bytes=None):
    """roofline object of the f16 head / fused f16 frame kernel: ALGORITHMIC FLOP/s (46 368 FLOP per marched sample, the network's own
    count) against the dense f16 MFMA peak -- `frac` is that quotient, nothing else.  The kernel's matrix work is small (59
    v_mfma_f32_16x16x32_f16 per 16-row slice); what holds it is vector-instruction ISSUE (gathers' index arithmetic, conversions, the
    march): the PMC passes (profiles/r4_f16_head_pmc_summary.json, rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_VALU_MFMA_MOPS_F16 over this
    same command) give the instructions per slice, and with 4 issue cycles per wave64 VALU instruction and 8 per MFMA
    (MI355X_MICROARCH.md, cycle constants; per 16 sample rows = half a 32-sample slice) the `valu_issue` block prices that stream -- a labelled DIAGNOSTIC of where the time goes,
    not a roofline: fewer instructions raise it."""
    t = head_total_ms * 1e-3
    sps = samples * steps / t
    tflops = FLOP_PER_SAMPLE * sps / 1e12
    r = dict(bound="mfma", achieved=round(tflops, 2), peak=F16_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tflops / F16_MFMA_PEAK_TFLOPS, 4), traffic=None,
             kernel="lz_k_frame<1,S,ROWS> (march+f16 head+composite)" if fused else "lz_k_triplane_head_f16w",
             avg_launch_ms=round(head_total_ms / max(launches_with_work, 1), 5), avg_launch_ms_all=round(head_total_ms / max(n_launch, 1), 5),
             launches=n_launch, head_time_share=round(t / dt, 4), head_ms_per_step=round(head_total_ms / steps, 4), rows_per_frame=rows,
             samples_per_s=round(sps, 1), flop_per_sample=FLOP_PER_SAMPLE,
             matrix_pipe_busy_frac=round(F16_SLICE_MFMAS * F16_MFMA_CYCLES * (rows / 16) * steps / (t * 2.4e9 * 1024), 4))
    if fused and n_rays and packed_bytes:   # 24 B/ray in + ~68 B/ray out + the packed f16 weights every workgroup stages into its LDS once
        r["algorithmic_bytes_per_launch"] = round(92 * n_rays + min(256, (n_rays + 63) // 64) * packed_bytes)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", F16_PMC_SUMMARY)))
        k = next(kk for kk in pmc["SQ_INSTS_VALU"] if kk.startswith("lz_k_frame<1,")) if fused else "lz_k_triplane_head_f16w"
        n_mfma = pmc["SQ_INSTS_VALU_MFMA_MOPS_F16"][k]["avg_per_launch"] / 64.0       # MOPS counts 512-FLOP units: 64 per 32x32x16 MFMA
        slices = n_mfma / F16_SLICE_MFMAS
        valu = pmc["SQ_INSTS_VALU"][k]["avg_per_launch"] - n_mfma
        cyc = (4.0 * valu + 8.0 * n_mfma) / slices                                      # issue-port cycles per 16-row slice
        bound = 1024 * 2.4e9 / cyc * 16 * (samples / max(rows, 1))                       # samples/s the chip's 1024 SIMDs could issue at 2.4 GHz
        fetch, write = pmc.get("FETCH_SIZE", {}).get(k), pmc.get("WRITE_SIZE", {}).get(k)
        if fetch and write:   # KiB per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950
            r["traffic"] = round((2 * _pmc_per_launch(fetch) + _pmc_per_launch(write)) * 1024)
            r["traffic_unit"] = "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE of " + F16_PMC_SUMMARY + ")"
        r["valu_issue"] = dict(note="diagnostic, not the roofline: the kernel's own instruction stream priced at the nominal clock",
                               valu_insts_per_slice=round(valu / slices, 1), mfma_per_slice=F16_SLICE_MFMAS, issue_cycles_per_slice=round(cyc, 1),
                               samples_per_s_if_issue_port_saturated=round(bound, 1), fraction_of_that=round(sps / bound, 4),
                               vmem_reads_per_slice=round(pmc["SQ_INSTS_VMEM_RD"][k]["avg_per_launch"] / slices, 1),
                               lds_insts_per_slice=round(pmc["SQ_INSTS_LDS"][k]["avg_per_launch"] / slices, 1), pmc="profiles/" + F16_PMC_SUMMARY)
    except (OSError, KeyError, ValueError, ZeroDivisionError, StopIteration):
        r["note"] = "PMC summary absent: instruction counts not reported"
    return r


def side_legs(args, result, device, P, golden, sd, head, bits, bits_dev, job, image, samples_per_frame, make_job):
    """N = 1 only: everything reported beside the headline.  A leg that fails is recorded in `leg_errors`, never fatal."""
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    H = W = args.size
    N = H * W
    enc_a, ind, eye = job.cond
    rays_o, rays_d = job.sf.rays(job.pose, job.intr)
    pose = job.pose.cpu().numpy()

    def err(name, exc):
        result.setdefault("leg_errors", {})[name] = repr(exc)
        log(f"leg {name} failed: {exc!r}")

    def side_leg(h, budget_factor, n_step_cap, scene_bits=None, size=None, steps=None, mode="loop"):
        """same frame (or `size`^2 with `scene_bits`), K timed steps after 2 warm-ups, with another head precision, schedule or mode"""
        j2 = make_job("frame", args.tiles, budget_factor, n_step_cap, h, mode=mode)
        if scene_bits is not None:
            j2.r.bitfield = scene_bits
        if size is not None:
            from lzzx_nerf_amd import dist as D
            from lzzx_nerf_amd.synthetic import synthetic_camera
            j2.H = j2.W = size
            j2.intr = synthetic_camera(size, size)[1]
            j2.sf = D.ShardedFrame(size, size, 0, 1, device=device)
        k = steps or args.steps
        d2, hms, o2, _ = timed(j2, k, 2, 1, device)
        s2 = o2["state"].cpu().numpy()
        return dict(mode=mode, schedule=f"one persistent kernel, slots refilled on the fly, cap = {args.cap}" if mode == "fused" else
                    f"n_step = max(min({budget_factor} * N // n_alive, {n_step_cap}), 1)", value=round(int(s2[5]) * k / d2, 1),
                    unit="samples/s", ms_per_step=round(d2 / k * 1e3, 4), rays_per_s=round(j2.sf.n_local * k / d2, 1),
                    samples_per_frame=int(s2[5]), iterations_per_frame=int(s2[6]), rows_per_frame=int(s2[72]),
                    head_ms_per_step=round(float(np.sum(hms)) / k, 4)), o2["image"].clone(), float(np.sum(hms)), s2

    log("leg: if (args.budget_factor, args.n_step_cap) != REF_SCHEDULE and not args.")
    if args.mode == "fused" and not args.no_fat_schedule:
        try:
            # the multi-launch loop under --budget-factor / --n-step-cap (round 1's headline path): same pixels
            leg, limg, lms, lst = side_leg(head, args.budget_factor, args.n_step_cap)
            leg["image_equal_to_headline"] = bool(torch.equal(limg, image))
            leg["samples_equal_to_headline"] = bool(int(lst[5]) == samples_per_frame)
            if args.precision == "f32":
                leg["head_frac"] = round(FLOP_PER_SAMPLE * int(lst[5]) * args.steps / (lms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)
            result["loop_mode"] = leg
        except Exception as exc:
            err("loop_mode", exc)
    if ((args.budget_factor, args.n_step_cap) != REF_SCHEDULE or args.mode == "fused") and not args.no_fat_schedule:
        try:
            # the same frame under the reference's own iteration schedule (1 x N sample rows per iteration, <= 8 steps per ray): more,
            # thinner launches; pixels and per-ray sample counts must be identical
            leg, fimg, fms, fst = side_leg(head, *REF_SCHEDULE)
            leg["schedule"] += " (the reference's, renderer.py:513)"
            leg["image_equal_to_headline_schedule"] = bool(torch.equal(fimg, image))
            leg["samples_equal_to_headline_schedule"] = bool(int(fst[5]) == samples_per_frame)
            if args.precision == "f32":
                leg["head_frac"] = round(FLOP_PER_SAMPLE * int(fst[5]) * args.steps / (fms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)
                leg["head_issued_frac"] = round(ISSUED_FLOP_PER_ROW * int(fst[72]) * args.steps / (fms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)
            result["reference_schedule"] = leg
        except Exception as exc:
            err("reference_schedule", exc)
    if args.mode == "fused" and not args.no_fat_schedule:
        try:
            # the reference's DEPLOYED cap (HubertInferenceMQ.py:69, train.py:35: max_steps = 16) on the head-like ellipsoid scene: dt_min =
            # dt_max, 16 steps cross 0.43 units, so the cap binds on most foreground rays and the reference hands each of them C_eff =
            # sum of n_step > 16 samples.  The fused frame against the multi-launch loop under the reference's schedule: pixels and counts.
            from lzzx_nerf_amd.synthetic import ellipsoid_bitfield_device
            ebits16 = ellipsoid_bitfield_device(device)[0]
            keep_ms = args.max_steps
            args.max_steps = 16
            h16 = head if args.precision == "f16" else FusedTriplaneHead(sd, bound=1.0, device=device, precision="f16")
            try:
                jf = make_job("frame", args.tiles, 1, 8, head, mode="fused")
                jl = make_job("frame", args.tiles, 1, 8, head, mode="loop")
                j16 = make_job("frame", args.tiles, 1, 8, h16, mode="fused")      # the deployed ARITHMETIC too: opt.fp16 (HubertInferenceMQ.py:53)
            finally:
                args.max_steps = keep_ms
            jf.r.bitfield = jl.r.bitfield = j16.r.bitfield = ebits16
            k16 = args.steps
            df, _, of_, _ = timed(jf, k16, 2, 1, device)
            img_f, st_f = of_["image"].clone(), of_["state"].cpu().numpy()
            dl, _, ol_, _ = timed(jl, k16, 2, 1, device)
            st_l = ol_["state"].cpu().numpy()
            cf = jf.r.render(*jf.sf.rays(jf.pose, jf.intr), *jf.cond, max_steps=16, count_samples=True)["ray_counts"].clone()
            cl = jl.r.render(*jl.sf.rays(jl.pose, jl.intr), *jl.cond, max_steps=16, count_samples=True)["ray_counts"]
            d16, _, o16_, _ = timed(j16, k16, 2, 1, device)
            result["deployed_max_steps_16"] = dict(
                f16_ms_per_step=round(d16 / k16 * 1e3, 4), f16_max_abs_diff_vs_f32_image=float((o16_["image"] - img_f).abs().max()),
                workload=f"{H}x{W} frame, ellipsoid occupancy, max_steps 16 (the reference's deployed value), cap = {args.cap}",
                ms_per_step=round(df / k16 * 1e3, 4), composited_samples_per_frame=int(st_f[5]), value=round(int(st_f[5]) * k16 / df, 1), unit="samples/s",
                c_eff=int(st_f[10]), reference_loop_iterations=int(st_f[11]), rays_continued_past_max_steps=int(st_f[9]),
                loop_mode_reference_schedule_ms_per_step=round(dl / k16 * 1e3, 4), loop_mode_iterations=int(st_l[6]),
                image_equal_to_reference_schedule=bool(torch.equal(img_f, ol_["image"])),
                ray_counts_equal_to_reference_schedule=bool(torch.equal(cf, cl)), max_ray_count=int(cf.max()))
            del jf, jl, j16
        except Exception as exc:
            err("deployed_max_steps_16", exc)
    if args.precision == "f32" and args.mode == "fused" and not args.no_fat_schedule:
        try:
            # the same frame with geo = Wg s2 folded into color_net.0 at pack time (head.py fold_geo, precision 2): 297 instead of 361
            # MFMAs per slice; sigma, sample counts, weights and depth are the headline's bit for bit, rgb moves by the reassociation
            hf = FusedTriplaneHead(sd, bound=1.0, device=device, fold_geo=True)
            leg, imgf, fms, fst = side_leg(hf, args.budget_factor, args.n_step_cap, mode="fused")
            leg["max_abs_diff_vs_headline_image"] = float((imgf - image).abs().max())
            leg["samples_equal_to_headline"] = bool(int(fst[5]) == samples_per_frame)
            # FLOP_PER_SAMPLE is the reference network's count; the folded kernel issues 8 192 FLOP per sample fewer
            leg["frac_of_f32_mfma_peak_algorithmic"] = round(FLOP_PER_SAMPLE * int(fst[5]) * args.steps / (fms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)
            leg["frac_of_f32_mfma_peak_executed"] = round((FLOP_PER_SAMPLE - 8192) * int(fst[5]) * args.steps / (fms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)
            result["folded_geo"] = leg
            del hf
        except Exception as exc:
            err("folded_geo", exc)
    h16 = None
    log("leg: if args.precision == 'f32' and not args.no_fp16_leg:")
    if args.precision == "f32" and not args.no_fp16_leg:
        try:
            # the reference's opt.fp16 arithmetic (torch autocast) on the f16 matrix cores: a different rounding sequence, so it is
            # reported beside the bit-exact f32 headline, with its distance from the f32 image
            h16 = FusedTriplaneHead(sd, bound=1.0, device=device, precision="f16")
            leg, img16, hms16, s16 = side_leg(h16, args.budget_factor, args.n_step_cap, mode=args.mode)
            leg8, img16b, _, _ = side_leg(h16, *REF_SCHEDULE)
            if args.mode == "fused":
                legl, img16l, _, _ = side_leg(h16, args.budget_factor, args.n_step_cap)
                leg.update(loop_mode_value=legl["value"], loop_mode_ms_per_step=legl["ms_per_step"], loop_mode_image_equal=bool(torch.equal(img16, img16l)))
            diff = (img16 - image).double()
            mse16 = float((diff ** 2).mean())
            leg.update(dtype="f16 (f32 accumulate, torch-autocast rounding)", kernel="lz_k_triplane_head_f16w",
                       max_abs_diff_vs_f32_image=float(diff.abs().max()), psnr_vs_f32_image_db=round(-10 * np.log10(max(mse16, 1e-300)), 2),
                       reference_schedule_value=leg8["value"], reference_schedule_ms_per_step=leg8["ms_per_step"],
                       reference_schedule_image_equal=bool(torch.equal(img16, img16b)),
                       roofline=f16_head_roofline(int(s16[5]), int(s16[72]), args.steps, hms16, int(s16[6]) * args.steps, int(s16[6]) * args.steps,
                                                  leg["ms_per_step"] * 1e-3 * args.steps, fused=args.mode == "fused", n_rays=N,
                                                  packed_bytes=h16.packed.numel() * h16.packed.element_size()))
            result["fp16_head"] = leg
        except Exception as exc:
            err("fp16_head", exc)
    log("leg: if not args.no_cfg5:")
    if not args.no_cfg5:
        try:
            # BASELINE cfg5: 1024 x 1024, ellipsoid occupancy (2.9 % of the cells: skipping + on-device compaction), fp16 MLP on MFMA;
            # few rays are alive in a sparse scene, so the reference's rule (1, 8) is also the best schedule here
            from lzzx_nerf_amd.synthetic import ellipsoid_bitfield_device
            ebits = ellipsoid_bitfield_device(device)[0]
            if h16 is None:
                h16 = FusedTriplaneHead(sd, bound=1.0, device=device, precision="f16")
            k5 = max(3, args.steps // 2)
            leg16, img5h, _, _ = side_leg(h16, *REF_SCHEDULE, scene_bits=ebits, size=1024, steps=k5, mode=args.mode)
            leg32, img5, _, _ = side_leg(head, *REF_SCHEDULE, scene_bits=ebits, size=1024, steps=k5, mode=args.mode) if args.precision == "f32" else (None, None, None, None)
            if args.mode == "fused":
                l5, i5, _, _ = side_leg(h16, *REF_SCHEDULE, scene_bits=ebits, size=1024, steps=k5)
                leg16.update(loop_mode_value=l5["value"], loop_mode_ms_per_step=l5["ms_per_step"], loop_mode_image_equal=bool(torch.equal(img5h, i5)))
            leg16.update(workload="cfg5: 1024x1024 frame, ellipsoid occupancy (2.9 % of cells), march_rays with on-device compaction, "
                                  "f16 MLP on MFMA (torch-autocast rounding)", dtype="f16", occupancy_fraction=round(float(np.unpackbits(ebits.cpu().numpy()).mean()), 4))
            if leg32 is not None:
                d5 = (img5h - img5).double()
                leg16.update(f32_head=dict(value=leg32["value"], ms_per_step=leg32["ms_per_step"], rays_per_s=leg32["rays_per_s"]),
                             max_abs_diff_vs_f32_image=float(d5.abs().max()),
                             psnr_vs_f32_image_db=round(-10 * np.log10(max(float((d5 ** 2).mean()), 1e-300)), 2))
            result["cfg5_1024_ellipsoid_f16"] = leg16
        except Exception as exc:
            err("cfg5_1024_ellipsoid_f16", exc)
    del h16
    log("leg: if not args.no_dense192:")
    if not args.no_dense192:
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
            err("dense192", exc)
    log("leg: cfg2 render")
    if not args.no_grid_roofline:
        try:
            # BASELINE cfg2 end to end: 256 x 256 rays, max_steps 128, generic hash-grid NeRF (D3 L16 C2 T19) through the operator API
            # with the reference's own inference loop (host sync per iteration and all) -- what a caller of the drop-in operators gets
            from lzzx_nerf_amd.synthetic import GenericHashgridNeRF, synthetic_camera as cam2
            from lzzx_nerf_amd.utils import frame_rays as fr2
            pose2, intr2 = cam2(256, 256)
            ro2, rd2 = fr2(torch.from_numpy(np.ascontiguousarray(pose2)).to(device), intr2, 256, 256)
            aabb2 = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device=device)
            bits2 = torch.full((128 ** 3 // 8,), 255, dtype=torch.uint8, device=device)
            legs2 = {}

            def time2(f):
                for _ in range(2):
                    o = f()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    o = f()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / 5 * 1e3, o
            for tag, half in (("f32_tables", False), ("f16_tables", True)):
                g2 = GenericHashgridNeRF(device, half_tables=half)
                ms2, out2 = time2(lambda: g2.render(ro2, rd2, aabb2, bits2, max_steps=128))
                leg = dict(reference_loop=dict(ms_per_frame=round(ms2, 3), rays_per_s=round(65536 / ms2 * 1e3, 1), sample_rows_per_frame=int(out2[3]),
                                               iterations_per_frame=int(out2[4]), note="renderer.py:495-561 on the operators: boolean-mask compaction, a host sync per iteration"))
                img_ref = out2[0].clone()
                from lzzx_nerf_amd.renderer import NetworkRenderer
                for name, graph in (("device_loop", False), ("device_loop_hipgraph", True)):
                    nr = NetworkRenderer(lambda x, d: g2.net(x, d, 1.0), bits2, bound=1.0, aabb=aabb2, graph=graph)
                    ms3, o3 = time2(lambda: nr.render(ro2, rd2, max_steps=128))
                    st3 = o3["state"].cpu().numpy()
                    leg[name] = dict(ms_per_frame=round(ms3, 3), rays_per_s=round(65536 / ms3 * 1e3, 1), samples_per_frame=int(st3[5]),
                                     samples_per_s=round(int(st3[5]) / ms3 * 1e3, 1), iterations_per_frame=int(st3[6]), rows_evaluated_per_frame=int(st3[6]) * 4 * 65536,
                                     image_equal_to_reference_loop=bool(torch.equal(o3["image"], img_ref)))
                    del nr
                # the fused path (lzzx_nerf_amd/ngp.py): tiled level-major gather + one MFMA head kernel + composite, 4 launches per iteration
                from lzzx_nerf_amd.ngp import FusedHashgridNeRF, HashgridRenderer
                fnet = FusedHashgridNeRF(g2.enc, g2.sigma_net, g2.color_net, half_tables=half)
                best = None
                for sched in ((8, 8), (4, 4), (8, 16), (1, 8)):
                    hr = HashgridRenderer(fnet, bits2, bound=1.0, aabb=aabb2, budget_factor=sched[0], n_step_cap=sched[1])
                    ms4, o4 = time2(lambda: hr.render(ro2, rd2, max_steps=128))
                    st4 = o4["state"].cpu().numpy()
                    # rays that still had samples to take at max_steps: only THEY see the schedule (the reference hands them its C_eff); 0 = the
                    # frame is the reference schedule's under any schedule
                    at_cap = int((hr.render(ro2, rd2, max_steps=128, count_samples=True)["ray_counts"] >= 128).sum())
                    d4 = dict(rays_at_max_steps=at_cap, ms_per_frame=round(ms4, 3), rays_per_s=round(65536 / ms4 * 1e3, 1), samples_per_frame=int(st4[5]),
                              samples_per_s=round(int(st4[5]) / ms4 * 1e3, 1), iterations_per_frame=int(st4[6]), sample_rows_per_frame=int(st4[72]),
                              schedule=f"n_step = max(min({sched[0]} * N // n_alive, {sched[1]}), 1)" + (" (the reference's)" if sched == (1, 8) else ""),
                              max_abs_diff_vs_reference_loop_image=float((o4["image"] - img_ref).abs().max()))
                    leg.setdefault("fused_schedules", {})["%dx%d" % sched] = d4
                    if best is None or ms4 < best[0]:
                        best = (ms4, d4)
                    del hr
                # like for like: `fused` is the REFERENCE schedule (1, 8) -- the reference's cap semantics on rays that reach max_steps; the faster
                # schedules (other pixels on such rays, identical ones elsewhere) are listed in fused_schedules, the fastest named here
                ref_leg = leg["fused_schedules"]["1x8"]
                leg["fused"] = dict(ref_leg, note="lz_ngp_loop_run under the reference's schedule (1, 8): march -> level-major gather (tiled, never untiled) -> "
                                    "lz_k_ngp_head (both MLPs + SH + activations on v_mfma_f32_16x16x4_f32) -> composite; pixels differ from the "
                                    "operator-API network only by the Linear layers' summation order",
                                    fastest_schedule=best[1]["schedule"], fastest_schedule_ms_per_frame=best[1]["ms_per_frame"],
                                    fastest_schedule_rays_at_max_steps=best[1]["rays_at_max_steps"],
                                    differs_from_reference_loop_image=bool(ref_leg["max_abs_diff_vs_reference_loop_image"] > 1e-4))
                best = (ref_leg["ms_per_frame"], ref_leg)
                # roofline of the leg: the gather's algorithmic bytes (1 164 B per sample f32 tables, 588 B f16) over the whole frame time
                per = 588 if half else 1164
                leg["fused"]["roofline"] = dict(bound="hbm", unit="GB/s", peak=8000.0, achieved=round(per * best[1]["sample_rows_per_frame"] / best[0] / 1e6, 1),
                                                frac=round(per * best[1]["sample_rows_per_frame"] / best[0] / 1e6 / 8000.0, 4),
                                                note="algorithmic gather bytes of the rows evaluated / whole-frame time (march, head and composite included)")
                del fnet
                legs2[tag] = leg
                del g2
            result["cfg2_hashgrid_render"] = dict(
                workload="cfg2: 256x256 rays, max_steps 128, all-ones occupancy, get_encoder('hashgrid') defaults (D3 L16 C2 T2^19) + SH(4) + "
                         "bias-free MLPs 32-64-16 / 31-64-3 (MFMA Linear kernels); reference_loop = the reference's inference loop on the operator API, "
                         "device_loop = renderer.NetworkRenderer (loop state on the device, the network on the whole 4 N row budget per iteration, no "
                         "host round trip), device_loop_hipgraph = the same with two iterations captured as one hipGraph and replayed", **legs2)
        except Exception as exc:
            err("cfg2_hashgrid_render", exc)
    log("leg: if not args.no_grid_roofline:")
    if not args.no_grid_roofline:
        try:
            result["roofline_gridencoder"] = grid_roofline(device)
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["roofline_gridencoder"] = repr(exc)
    log("leg: if not args.no_train:")
    if not args.no_train:
        try:
            result["train_step"] = train_bench(args, device, P, golden, bits)
            torch.cuda.empty_cache()
            if args.train_mlp == "fused" and not args.train_recompute and args.train_records == "f32" and args.train_forward == "f32" and args.train_backward == "f32":
                # the same step with the weight-gradient operands in half (what the reference's autocast mode feeds its dW GEMMs)
                import copy
                a16 = copy.copy(args)
                a16.train_records = "f16"
                result["train_step_f16_records"] = train_bench(a16, device, P, golden, bits)
                torch.cuda.empty_cache()
                # and with the forward itself in the reference's autocast arithmetic (its usual `-O` training mode)
                a16.train_forward = "f16"
                result["train_step_f16_forward"] = train_bench(a16, device, P, golden, bits)
                torch.cuda.empty_cache()
                # and the data gradient on the f16 matrix cores too: the whole step in autocast arithmetic
                a16.train_backward = "f16"
                result["train_step_f16"] = train_bench(a16, device, P, golden, bits)
                torch.cuda.empty_cache()
            # BASELINE cfg3's second size: every ray of the 512 x 512 frame (N = 262 144; 24 M samples, 80 GB of per-sample records + state)
            result["train_step_full_frame"] = train_bench(args, device, P, golden, bits, n_rays=H * W)
            torch.cuda.empty_cache()
            if args.train_mlp == "fused" and not args.train_recompute and (args.train_records, args.train_forward, args.train_backward) == ("f32", "f32", "f32"):
                import copy
                a16 = copy.copy(args)
                a16.train_records = a16.train_forward = a16.train_backward = "f16"
                result["train_step_full_frame_f16"] = train_bench(a16, device, P, golden, bits, n_rays=H * W)
                torch.cuda.empty_cache()
        except Exception as exc:   # an optional leg must never take the headline line down
            err("train_step", exc)
    log("leg: if not args.no_occupancy and args.precision == 'f32':")
    if not args.no_occupancy and args.precision == "f32":
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
    log("leg: if not args.no_occupancy:")
    if not args.no_occupancy:
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
                                          note="all pixels queried (no 2-D occupancy mask); 5.4 kMAC per pixel on v_mfma_f32_16x16x4_f32, 16 pixels per wave pass")
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
            del aenc
            # the whole frame the way run_cuda_for_inference orders it (renderer.py:406-570): rays -> audio window -> enc_a -> torso
            # background -> head (the fused frame kernel over that background) -> blended frame + the video pipe's RGB24
            from lzzx_nerf_amd.pipeline import TalkingHeadFrame
            from lzzx_nerf_amd.utils import frame_rays as fr3
            full_sd = dict(sd)
            full_sd.update(sdt)
            full_sd.update(sda)
            lin1 = torch.linspace(-1, 1, H, device=device)
            bgc = torch.stack(torch.meshgrid(lin1, lin1, indexing="xy"), -1).reshape(-1, 2).contiguous()
            pose_d = job.pose
            legs3 = {}
            for prec in ("f32", "f16"):
                thf = TalkingHeadFrame(full_sd, bits_dev, bound=1.0, precision=prec, device=device, mode="fused")

                def whole():
                    ro3, rd3 = fr3(pose_d, job.intr, H, W)
                    return thf.render(ro3, rd3, auds, eye=eye, ind_code=ind, bg_coords=bgc, poses=pose_d[None], ind_code_torso=indt,
                                      bg_color=1.0, max_steps=args.max_steps, rgb24=True)
                for _ in range(3):
                    whole()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    o3 = whole()
                torch.cuda.synchronize()
                ms3 = (time.perf_counter() - t0) / 10 * 1e3
                legs3[prec] = dict(ms_per_frame=round(ms3, 4), frames_per_s=round(1e3 / ms3, 1), samples_per_frame=int(o3["state"][5]))
                del thf
            result["talking_head_frame"] = dict(
                workload=f"{H}x{W} frame, max_steps {args.max_steps}: ray generation + encode_audio (8 HuBERT windows) + torso branch on every pixel + "
                         "triplane head over the torso background (fused frame kernel) + blend + RGB24, random torso / audio weights", **legs3)
            del auds, bgc
        except Exception as exc:   # an optional leg must never take the headline line down
            result.setdefault("leg_errors", {})["torso_audio"] = repr(exc)
    # ---- CPU baseline (rank 0, N = 1): bounded, next to the GPU numbers; also PSNR / sample-count parity against the pinned checker ----
    log("leg: if not args.no_cpu_baseline:")
    if not args.no_cpu_baseline:
        try:
            cnt = job.r.render(rays_o, rays_d, enc_a, ind, eye, dt_gamma=1 / 256, max_steps=args.max_steps, T_thresh=1e-4,
                               count_samples=True)["ray_counts"].cpu().numpy()
            img_np = image.cpu().numpy()
            base, parity = cpu_baseline(args, P, golden, bits, lambda sel: img_np[sel], lambda sel: cnt[sel])
            result["cpu_baseline"] = base
            result.update(parity)
        except Exception as exc:   # reported, never fatal for the line
            err("cpu_baseline", exc)
