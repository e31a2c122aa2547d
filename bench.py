#!/usr/bin/env python3
"""Headline benchmark: 512x512 triplane-head inference render at max_steps = 192 on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process spawns `python -m torch.distributed.run --nproc-per-node N` on itself
BEFORE touching the GPU and exits with the children's code; under torch.distributed.run it is one rank per GPU over RCCL.

A "step" renders one full synthetic frame through the whole hot path: ray generation -> near/far -> [march -> fused head -> composite ->
compaction] until every ray is done -> blend (-> tile all-gather), inputs resident in HBM, no host sync inside the frame.  Workload
(SURVEY 8d): camera at (0,0,-3.35) looking down +z, fovy 21.24 deg, bound 1, aabb [-1,-.5,-1,1,.5,1], dt_gamma 1/256, T_thresh 1e-4,
all-ones occupancy grid (dense: nothing is skipped), triplane tables U(-1,1), MLP weights = the reference's torch init under seed 0
(tests/golden fixture), enc_a ~ N(0,1), eye 0.25.  Data is synthetic.

N > 1, `--shard frame` (default; BASELINE cfg4 as north_star states it, STRONG scaling): ONE 512x512 frame per step whose rays are
sharded over the N ranks in row tiles -- `--tiles interleaved` (default: 8-row stripes dealt round-robin, balances the expensive
middle of the frame) or `contiguous` (rank g renders rays [g N/G, (g+1) N/G)) --, every rank renders its tile with a full model
replica and ONE RCCL all-gather per frame makes every rank hold the frame (overlapped with the next frame's march).  The other
tiling and the round-1 clip mode (`--shard clip`: N consecutive frames of a talking-head clip, rank r renders frame r, weak scaling)
are timed in the same run and reported as labelled legs.

Iteration schedule: every loop iteration marches n_step = max(min(F * N // n_alive, C), 1) samples per alive ray.  The reference uses
F = 1, C = 8 (renderer.py:513), i.e. N sample rows per iteration and thin launches while most rays are alive; pixels do not depend on F
and C (each ray marches the same sample sequence and compositing resumes exactly), so the headline runs F = C = 4 and the
`reference_schedule` leg times F = 1, C = 8 on the same frame and checks that the image and the sample count are identical.

Prints one JSON line on rank 0.  `value` = marched samples (delta != 0) per second over all ranks.
"""
import os
os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # cpu_baseline leg: two OpenMP runtimes (torch's, the checker's) must not spin against each other
import argparse  # noqa: E402
import ctypes as C
import json
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.bench_contract import compact  # noqa: E402
from tools.bench_legs import (F32_MFMA_PEAK_TFLOPS, FLOP_PER_SAMPLE, ISSUED_FLOP_PER_ROW, PMC_SUMMARY, FrameJob, _pmc_per_launch,  # noqa: E402
                              f16_head_roofline, frame_stats, log, side_legs, timed, train_bench)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run on this same script, one rank per GPU.  Runs before
    anything touches the GPU (an exec / fork from a process that has initialised HIP is not allowed on the pool); the ranks' stdout
    (rank 0 prints the JSON line) and exit code pass through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget-factor", type=int, default=4,
                    help="sample rows per iteration = factor x rays (reference: 1; pixels and sample counts do not depend on it)")
    ap.add_argument("--n-step-cap", type=int, default=4, help="max samples per ray per iteration (reference: 8)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"],
                    help="f16 = the reference's opt.fp16 / autocast arithmetic on the f16 matrix cores (not bit-exact vs the f32 checker)")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--train-only", action="store_true", help="time only the cfg3 training step (tools/profile_train.sh) and print it")
    ap.add_argument("--train-rays", type=int, default=65536)
    ap.add_argument("--train-records", default="f32", choices=["f32", "f16"],
                    help="fused training head: precision of the per-sample records the weight gradients are reduced from (f16 = the operand "
                         "rounding of the reference's autocast dW GEMMs; forward, data gradient and accumulation stay f32)")
    ap.add_argument("--train-forward", default="f32", choices=["f32", "f16"],
                    help="fused training head: f16 = the forward in the reference's autocast arithmetic on the f16 matrix cores (implies f16 records)")
    ap.add_argument("--train-backward", default="f32", choices=["f32", "f16"],
                    help="fused training head: f16 = the data-gradient products on the f16 matrix cores (autocast's half backward; implies f16 records)")
    ap.add_argument("--train-keep-records", action="store_true",
                    help="all-f16 training leg: the recorded pair of rounds 2-4 (1 216 B of record + state per sample) instead of the recomputing one (80 B)")
    ap.add_argument("--train-layout", default="step", choices=["step", "ray"],
                    help="training legs: sample rows of march_rays_train -- 'step' = step-major groups of 64 neighbouring rays (round 5, "
                         "lz_march_rays_train_grouped), 'ray' = the reference's ray-major rows")
    ap.add_argument("--train-recompute", action="store_true",
                    help="fused training head with record=False: the backward recomputes the forward instead of reading what it recorded")
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
    ap.add_argument("--shard", default="frame", choices=["frame", "clip"],
                    help="N > 1: 'frame' = one frame ray-sharded N ways + one all-gather (cfg4, strong scaling); 'clip' = rank r renders "
                         "frame r of an N-frame clip (weak scaling)")
    ap.add_argument("--tiles", default="interleaved", choices=["interleaved", "contiguous"], help="--shard frame: how rows are dealt to ranks")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="single GPU: render rank 0's tile of a frame sharded this many ways (no collective) -- what one rank of cfg4 does")
    ap.add_argument("--mode", default="fused", choices=["fused", "loop"],
                    help="fused: the frame as one persistent kernel (csrc/lz_frame.hip; the loop under the schedule n_step = 1, same pixels); "
                         "loop: march / head / composite launches per iteration under --budget-factor / --n-step-cap")
    ap.add_argument("--steps-per-pass", type=int, default=0, help="fused mode: samples per ray and pass (0 = auto by ray count)")
    ap.add_argument("--cap", default="reference", choices=["reference", "per_ray"],
                    help="fused mode, rays still alive at max_steps: 'reference' = the reference loop's frame-wide count C_eff = sum of its n_step "
                         "(renderer.py:503-548; ranks of a sharded frame sum a small histogram between the kernel's two phases), "
                         "'per_ray' = stop at ceil(max_steps / S) * S (no exchange; not the reference's pixels on such rays)")
    ap.add_argument("--no-side-legs", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--detail", default="", help="where the full result (every side leg) is written; default bench_detail.json (under gpurun_out/ if present)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-sample", action="store_true",
                    help="time the CPU baseline on every 4th pixel of the cfg3 frame (median of 3) instead of the whole frame (median of 5)")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip the 8 extra 2M-row head launches that measure the sustained shader clock (tools/profile_bench.sh: the kernel "
                         "trace then holds only warm-up + timed frames, so its average head duration is directly bench's avg_launch_ms_all)")
    ap.add_argument("--no-grid-roofline", action="store_true")
    ap.add_argument("--no-reference-schedule", "--no-fat-schedule", dest="no_fat_schedule", action="store_true",
                    help="skip the leg that re-renders the frame under the reference's own iteration schedule (1 x N rows, <= 8 steps)")
    ap.add_argument("--no-fp16-leg", action="store_true")
    ap.add_argument("--no-occupancy", action="store_true")
    ap.add_argument("--no-dense192", action="store_true")
    ap.add_argument("--no-cfg5", action="store_true")
    ap.add_argument("--gather-via", default="collective", choices=["collective", "peer"],
                    help="N > 1, --shard frame: 'collective' = one RCCL all_gather_into_tensor per frame; 'peer' = every rank copies its tile "
                         "straight into each peer's frame buffer (one xGMI hop per tile, IPC-mapped buffers, device-side flag wait)")
    ap.add_argument("--verify-frames", type=int, default=0, metavar="K",
                    help="N > 1, --shard frame: after the timed region run K more steps and compare the CONTENTS of every gathered frame on every "
                         "rank with the frame of the timed region's last step, with no barrier between the steps (the check --gather-via peer "
                         "needs: a stale cache line or a torn hand-off shows up as a mismatch count in `verify_frames`)")
    ap.add_argument("--gather", default="f32", choices=["f32", "rgb24"],
                    help="what the per-step all-gather moves: f32 RGB tiles, or the video pipe's RGB24 quantised on device (4x fewer bytes)")
    args = ap.parse_args()
    if args.no_side_legs:
        args.no_train = args.no_cpu_baseline = args.no_grid_roofline = args.no_fat_schedule = args.no_fp16_leg = True
        args.no_occupancy = args.no_dense192 = args.no_cfg5 = True
    return args


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))          # nothing above this line touches the GPU

    from lzzx_nerf_amd import _lib, dist as D
    rank, world = D.init_from_env()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world == 1:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())
    assert _lib.load().lz_device_ok() == 1, "bench needs a gfx950 device; there is no fallback path"

    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    from lzzx_nerf_amd.synthetic import ellipsoid_bitfield_device, load_golden, make_params, ones_bitfield, orbit_pose, synthetic_camera

    golden = load_golden()
    P = make_params(golden)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    sd = {k: torch.from_numpy(v) for k, v in P.items()}
    head = FusedTriplaneHead(sd, bound=1.0, device=device, precision=args.precision)
    bits_dev = dev(ones_bitfield()) if args.scene == "ones" else ellipsoid_bitfield_device(device)[0]
    bits = bits_dev.cpu().numpy()
    H = W = args.size
    N = H * W
    _, intr = synthetic_camera(H, W)
    enc_a0, ind, eye = dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"])

    def make_job(shard, tiles, budget_factor=None, n_step_cap=None, h=None, shard_of=0, mode=None):
        r = TriplaneRenderer(h or head, bits_dev, bound=1.0, budget_factor=budget_factor or args.budget_factor,
                             n_step_cap=n_step_cap or args.n_step_cap, mode=mode or args.mode, cap=args.cap)
        r.steps_per_pass = args.steps_per_pass
        k = rank if (world > 1 and shard == "clip") else 0     # clip mode: frame `rank` of the clip, with its own audio feature
        enc_a = enc_a0
        if k > 0:
            enc_a = enc_a0 + 0.5 * torch.randn(enc_a0.shape, device=device, generator=torch.Generator(device=device).manual_seed(100 + k))
        return FrameJob(r, H, W, orbit_pose(k), intr, (enc_a, ind, eye), args.max_steps, rank, world, shard, tiles, args.gather, device, shard_of,
                        via=args.gather_via)

    log(f"rank {rank}/{world}: setup done")
    if args.train_only:
        print(json.dumps({"train_step": train_bench(args, device, P, golden, bits, rank, world)}))
        return
    # ---- headline ----
    job = make_job(args.shard, args.tiles, shard_of=args.shard_of)
    dt, head_ms, out, tiles = timed(job, args.steps, args.warmup, world, device)
    dt_own = job.dt_own
    samples_per_step, iters_per_frame, rows_per_step = frame_stats(out, world, device)
    log(f"headline: {dt / args.steps * 1e3:.3f} ms/step")
    frames_per_step = world if (world > 1 and args.shard == "clip") else 1
    rays_per_step = N * frames_per_step if args.shard_of <= 1 else job.n_rays
    value = samples_per_step * args.steps / dt
    image = out["image"].clone()
    probe = (C.c_uint64 * 2)()
    if not args.no_clock_probe and world == 1 and not args.no_side_legs:
        # sustained shader clock under the head's load: 8 back-to-back 2M-row launches, counters of one wave of the last one
        gp = torch.Generator(device=device).manual_seed(1)
        xs = torch.rand(1 << 21, 3, device=device, generator=gp) * 2 - 1
        ds = torch.nn.functional.normalize(torch.randn(1 << 21, 3, device=device, generator=gp), dim=-1)
        for _ in range(8):
            head.forward(xs, ds, enc_a0, ind, eye)
        _lib.call("lz_debug_head_clocks", probe)
        del xs, ds

    # ---- N > 1: the other tiling and the clip (weak-scaling) mode, same run, labelled ----
    multi = {}
    if world > 1 and not args.no_side_legs:
        for tag, shard, tl in (("tiles_" + ("contiguous" if args.tiles == "interleaved" else "interleaved"), "frame",
                                "contiguous" if args.tiles == "interleaved" else "interleaved"), ("clip_weak_scaling", "clip", args.tiles)):
            if shard == args.shard and tl == args.tiles:
                continue
            j2 = make_job(shard, tl)
            d2, _, o2, _ = timed(j2, args.steps, 2, world, device, timing=False)
            s2, it2, _ = frame_stats(o2, world, device)
            multi[tag] = dict(value=round(s2 * args.steps / d2, 1), unit="samples/s", ms_per_step=round(d2 / args.steps * 1e3, 4),
                              scaling="weak" if shard == "clip" else "strong", iterations_per_frame=it2,
                              frames_per_step=world if shard == "clip" else 1, rays_per_rank=j2.n_rays,
                              parallelism=(f"clip of {world} frames, rank r renders frame r, 1 all-gather/step" if shard == "clip" else
                                           f"one frame ray-sharded x{world} ({tl} row tiles), 1 all-gather/frame"))
            del j2
    train_dp = None
    if world > 1 and args.train_dp:   # every rank takes part: one gradient all-reduce per step
        train_dp = train_bench(args, device, P, golden, bits, rank, world)
    # per-rank tile times (every rank's own clock and its persistent-kernel time per step), for the first SCALE record to read load balance from
    rank_tile_ms = None
    if world > 1:
        mine_ms = torch.tensor([dt_own / args.steps * 1e3, float(np.sum(head_ms)) / max(args.steps, 1)], dtype=torch.float64, device=device)
        allms = [torch.zeros_like(mine_ms) for _ in range(world)]
        torch.distributed.all_gather(allms, mine_ms)
        rank_tile_ms = dict(step=[round(float(t[0]), 4) for t in allms], kernel=[round(float(t[1]), 4) for t in allms])
    # correctness of the gathered frame on every rank: the assembled tiles equal this rank's own tile where they overlap
    gather_ok = gather_equals_unsharded = verify = None
    if world > 1 and args.shard == "frame":
        job.sf.wait()
        if args.verify_frames > 0:
            # K more frames of the same pose, NO barrier between them: every rank compares every gathered frame, word for word, with the frame
            # the timed region ended on (whose own check against the unsharded reference loop follows below)
            want = tiles.clone()
            bad = 0
            for _ in range(args.verify_frames):
                _, t2 = job.step()
                job.sf.wait()
                bad += int(not torch.equal(t2, want))
            vb = torch.tensor([float(bad)], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(vb)
            verify = dict(steps=args.verify_frames, mismatching_frames_over_all_ranks=int(vb.item()), barrier_between_steps=False, via=args.gather_via)
        frame = job.sf.assemble(tiles)
        mine = frame[job.sf.pixels]
        own = out["image_rgb24"] if args.gather == "rgb24" else out["image"]
        ok = torch.tensor([float(torch.equal(mine, own)), float(frame.shape[0] == N)], device=device)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        gather_ok = bool(ok.min().item() == 1.0)
        if rank == 0 and args.gather == "f32" and args.mode == "fused" and args.cap == "reference":
            # ... and the assembled frame equals the UNSHARDED frame of the multi-launch loop under the reference's own schedule (rank 0
            # renders it alone): the cap histogram the ranks summed makes their tiles the reference's, also where max_steps binds
            from lzzx_nerf_amd.utils import frame_rays as _fr
            whole = TriplaneRenderer(head, bits_dev, bound=1.0, mode="loop")
            ro_w, rd_w = _fr(job.pose, job.intr, H, W)
            ref_w = whole.render(ro_w, rd_w, *job.cond, dt_gamma=1 / 256, max_steps=args.max_steps, T_thresh=1e-4)["image"]
            gather_equals_unsharded = bool(torch.equal(frame, ref_w))
            del whole
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return

    # ---- roofline of the dominant kernel (fused head): live HIP events from the timed steps of rank 0 ----
    head_total_ms = float(np.sum(head_ms))
    n_launch = len(head_ms)
    st0 = out["state"].cpu().numpy()
    my_samples, my_rows = int(st0[5]), int(st0[72])
    launches_with_work = int(st0[6]) * args.steps
    achieved_tflops = FLOP_PER_SAMPLE * my_samples * args.steps / (head_total_ms * 1e-3) / 1e12
    roofline = dict(bound="mfma", achieved=round(achieved_tflops, 3), peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(achieved_tflops / F32_MFMA_PEAK_TFLOPS, 4), traffic=None,
                    kernel="lz_k_frame<0,S,1> (march+head+composite)" if args.mode == "fused" else "lz_k_triplane_head<false>",
                    kernel_note="the phase-1 persistent launch of each frame (lz_timing pair); the cap's histogram / phase-2 launches are outside it and return at once on this frame (no ray reaches max_steps)",
                    avg_launch_ms=round(head_total_ms / max(launches_with_work, 1), 5),
                    avg_launch_ms_all=round(head_total_ms / max(n_launch, 1), 5), launches=n_launch,
                    launches_with_work=launches_with_work, flop_per_sample=FLOP_PER_SAMPLE,
                    head_time_share=round(head_total_ms * 1e-3 / dt, 4), head_ms_per_step=round(head_total_ms / args.steps, 4),
                    rows_per_frame=my_rows,
                    shader_clock_mhz_under_load=round(probe[0] / max(probe[1], 1) * 100.0, 1) if (args.precision == "f32" and probe[1]) else None,
                    issued_frac=round(ISSUED_FLOP_PER_ROW * my_rows * args.steps / (head_total_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4))
    if world > 1:
        roofline["note"] = "rank 0's head launches over rank 0's samples"
    if args.precision == "f16":
        roofline = f16_head_roofline(my_samples, my_rows, args.steps, head_total_ms, n_launch, launches_with_work, dt, fused=args.mode == "fused",
                                     n_rays=job.n_rays, packed_bytes=head.packed.numel() * head.packed.element_size())
    pmc_path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if os.path.exists(pmc_path) and args.precision == "f32":
        # HBM-side traffic of the head per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs, tools/profile_bench.sh): KiB per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
        # for gfx950 (128-B requests tallied at 64 B).  bench.py cannot collect counters itself; null when the summary is absent.
        try:
            pmc = json.load(open(pmc_path))
            k = "lz_k_frame<0, 1, 1>" if args.mode == "fused" else "lz_k_triplane_head<false>"
            roofline["traffic"] = round((2 * _pmc_per_launch(pmc["FETCH_SIZE"][k]) + _pmc_per_launch(pmc["WRITE_SIZE"][k])) * 1024)
            roofline["traffic_unit"] = f"bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, mean over the launches of profiles/{PMC_SUMMARY} without the largest)"
            # fused: 24 B/ray in + ~68 B/ray out, plus the packed weights every workgroup stages into its LDS once (256 x 94 KB)
            roofline["algorithmic_bytes_per_launch"] = (round(92 * job.n_rays + min(256, (job.n_rays + 63) // 64) * 94080) if args.mode == "fused" else
                                                        round(52 * my_rows * args.steps / max(n_launch, 1)))
        except (KeyError, ValueError):
            pass
    if world == 1:
        par = "single GPU" if args.shard_of <= 1 else f"rank 0's tile of a frame ray-sharded x{args.shard_of} ({args.tiles} row tiles), no collective"
    elif args.shard == "frame":
        how = "1 all-gather/frame" if args.gather_via == "collective" else "direct one-hop tile writes into every peer's frame buffer (no collective)"
        par = f"one frame ray-sharded x{world} ({args.tiles} row tiles), {how} ({args.gather} tiles), overlapped with the next frame"
    else:
        par = f"clip of {world} frames, rank r renders frame r, 1 all-gather/step ({args.gather} tiles)"
    result = {
        "metric": f"rendered samples/s ({H}x{W} triplane head, max_steps {args.max_steps})", "value": round(value, 1), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak" if (world > 1 and args.shard == "clip") else "strong", "vs_baseline": None,
        "dtype": "f32" if args.precision == "f32" else "f16 (f32 accumulate, torch-autocast rounding)", "data": "synthetic",
        "config": {"workload": f"cfg3/cfg4 inference: {H}x{W} frame, max_steps {args.max_steps}, triplane head (3 hash planes + audio/eye cond + SH4), "
                               f"occupancy={args.scene}; step = ray gen + near/far + march/head/composite + blend" + (" + tile all-gather" if world > 1 else ""),
                   "workload_detail": "3x D2/L12/C1 hash grid, bound 1, dt_gamma 1/256, T_thresh 1e-4 (SURVEY 8d)",
                   "rays_per_step": rays_per_step, "frames_per_step": frames_per_step, "rays_per_rank": job.n_rays,
                   "samples_per_step": samples_per_step, "iterations_per_frame": iters_per_frame,
                   "nominal_samples_per_frame": N * args.max_steps, "parallelism": par,
                   "mode": args.mode,
                   "schedule": ("one persistent kernel per frame (csrc/lz_frame.hip), finished ray slots refilled on the fly; rays still alive at "
                                "max_steps receive the reference loop's frame-wide count C_eff = sum of n_step (renderer.py:503-548: phase 1 to "
                                "max_steps, device-side replay of n_alive / n_step from a histogram, phase 2 to C_eff): same pixels and per-ray "
                                "counts as the multi-launch loop under the reference's schedule ('reference_schedule'; with the cap binding: "
                                "'deployed_max_steps_16')" if args.cap == "reference" else
                                "one persistent kernel per frame (csrc/lz_frame.hip): the loop under n_step = S with finished ray slots refilled "
                                "on the fly; a ray still alive at max_steps stops at ceil(max_steps / S) * S samples (--cap per_ray: NOT the "
                                "reference's count on such rays; none in this frame if 'reference_schedule.image_equal_to_headline_schedule')")
                               if args.mode == "fused" else
                               f"n_step = max(min({args.budget_factor} * N // n_alive, {args.n_step_cap}), 1)"
                               + (" (the reference's, renderer.py:513)" if (args.budget_factor, args.n_step_cap) == (1, 8) else
                                  " -- sample rows per iteration sized for 288 GB of HBM; the reference's rule is 1 x N rows, <= 8 steps "
                                  "(renderer.py:513): same pixels and per-ray sample counts, timed in 'reference_schedule'")},
        "rays_per_s": round(rays_per_step * args.steps / dt, 1),
        "samples_per_ray_mean": round(samples_per_step / max(rays_per_step, 1), 2),
        "roofline": roofline,
    }
    if gather_ok is not None:
        result["gathered_frame_ok"] = gather_ok
    if gather_equals_unsharded is not None:
        result["gathered_frame_equals_unsharded_reference_loop"] = gather_equals_unsharded
    if verify is not None:
        result["verify_frames"] = verify
    if rank_tile_ms is not None:
        result["rank_tile_ms"] = rank_tile_ms
    result.update(multi)
    if train_dp is not None:
        result["train_step"] = train_dp
    if world == 1 and not args.no_side_legs:
        side_legs(args, result, device, P, golden, sd, head, bits, bits_dev, job, image, samples_per_step, make_job)
    # the full result (side legs, prose) goes to a file; stdout carries ONE compact contract line (tools/bench_contract.py, < 8 KB)
    detail = args.detail or os.path.join("gpurun_out" if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "", "bench_detail.json")
    try:
        with open(os.path.join(ROOT, detail), "w") as f:
            json.dump(result, f, indent=1)
        log(f"detail: {detail} ({os.path.getsize(os.path.join(ROOT, detail))} bytes)")
    except OSError as exc:
        log(f"detail file not written: {exc!r}")
        detail = None
    line = json.dumps(compact(result, detail))
    log(f"contract line: {len(line)} bytes")
    print(line, flush=True)


if __name__ == "__main__":
    main()
