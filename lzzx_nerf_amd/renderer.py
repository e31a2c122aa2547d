"""Device-resident triplane renderer: the thin counterpart of NeRFRenderer.run_cuda_for_inference
(/root/reference/nerf_triplane/renderer.py:406-570) used by bench.py and the parity tests.

Same algorithm and iteration schedule as the reference loop -- per iteration: march n_step samples for every
alive ray, evaluate the head, composite, drop dead rays, n_step = max(min(N // n_alive, 8), 1), stop at
max_steps -- but (n_alive, n_step, step) live in device memory (lz_loop_state), compaction is an
order-preserving device scan, and the host never synchronises inside a frame: it enqueues iterations in chunks
and peeks at a pinned copy of the state between chunks to stop early.  The reference's own renderer keeps
working unmodified through the drop-in operators; this class is what removes its per-iteration host sync
(38 % of its loop time) and ~20 launches per iteration.
"""
import ctypes as C

import torch

from . import _lib
from ._util import call, ptr, stream
from .head import FusedTriplaneHead

_STATE_INTS = 80         # LZ_LOOP_STATE_INTS: lz_loop_state (8 ints) + 64 sample-count slots + statistics
STAT_ROWS = 72           # LZ_LOOP_STAT_ROWS: sample rows handed to the head (exhausted rows included)
MAX_RAYS_PER_PASS = 4096 * 256   # lz_loop_march sums at most 4096 workgroup counts
_N_SAMPLES_OFF = (74 + 2) * 4   # LZ_LOOP_NEXT + 2: n_samples of the iteration in flight (the head's `count`)


from .utils import frame_rays   # noqa: E402,F401  (full-image rays of one pose; utils.get_rays has the reference's signature)

import os as _os  # noqa: E402
_SPLIT_CAP = bool(_os.environ.get("LZ_FRAME_SPLIT_CAP"))

get_rays = frame_rays   # round-1 name, kept for callers of (pose, intrinsics, H, W) -> (rays_o, rays_d)


class _Buffers:
    def __init__(self, N, rows, device):
        f = dict(dtype=torch.float32, device=device)
        i = dict(dtype=torch.int32, device=device)
        self.N, self.rows = N, rows
        self.nears, self.fars = torch.empty(N, **f), torch.empty(N, **f)
        self.rays_alive = [torch.empty(N, **i), torch.empty(N, **i)]
        self.rays_t = torch.empty(N, **f)
        self.weights_sum, self.depth = torch.empty(N, **f), torch.empty(N, **f)
        self.image, self.out = torch.empty(N, 3, **f), torch.empty(N, 3, **f)
        self.amb_aud_sum, self.amb_eye_sum, self.unc_sum = torch.empty(N, **f), torch.empty(N, **f), torch.empty(N, **f)
        # n_alive * n_step <= rows = max(sample budget, N) always (renderer.py:513 with budget = N)
        M = rows
        self.xyzs, self.dirs, self.deltas = torch.empty(M, 3, **f), torch.empty(M, 3, **f), torch.empty(M, 2, **f)
        self.sigmas, self.rgbs = torch.empty(M, **f), torch.empty(M, 3, **f)
        self.amb_aud, self.amb_eye, self.unc = torch.empty(M, 1, **f), torch.empty(M, 1, **f), torch.empty(M, 1, **f)
        self.state = torch.zeros(_STATE_INTS, **i)
        self.workspace = torch.empty(4096, **i)
        self.ray_counts = torch.zeros(N, **i)
        self.state_ring = [torch.zeros(8, dtype=torch.int32).pin_memory() for _ in range(4)]


def _empty_result(device, count_samples=False, rgb24=False, ambient=True):
    """what the loops return for a batch of no rays (a rank whose tile of a small frame is empty): empty outputs, a finished state"""
    z = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=device)
    state = torch.zeros(1024, dtype=torch.int32, device=device)
    state[3] = 1
    res = dict(image=z(0, 3), image_raw=z(0, 3), weights_sum=z(0), depth=z(0), state=state, nears=z(0), fars=z(0))
    if ambient:
        res.update(amb_aud_sum=z(0), amb_eye_sum=z(0), uncertainty_sum=z(0))
    if count_samples:
        res["ray_counts"] = torch.empty(0, dtype=torch.int32, device=device)
    if rgb24:
        res["image_rgb24"] = torch.empty(0, 3, dtype=torch.uint8, device=device)
    return res


class TriplaneRenderer:
    """Inference renderer for one head / one occupancy grid.

    head:             FusedTriplaneHead
    density_bitfield: uint8 [cascade * grid_size^3 / 8] (cuda), as produced by raymarching.packbits
    """

    def __init__(self, head: FusedTriplaneHead, density_bitfield, bound=1.0, cascade=None, grid_size=128, aabb=None,
                 min_near=0.05, density_scale=1, budget_factor=1, n_step_cap=8, mode="loop", cap="reference"):
        """mode "loop": the reference's iteration structure, 3 launches per iteration (march, head, composite), schedule
        (budget_factor, n_step_cap).  mode "fused": the whole frame as one persistent kernel (csrc/lz_frame.hip) with on-the-fly refill of
        finished ray slots; `steps_per_pass` (0 = chosen from the ray count: 1 for a whole frame, up to 16 for small tiles) is its launch shape.
        Cap semantics: the reference tests `step < max_steps` once per ITERATION (renderer.py:503-548) with n_step = max(min(N // n_alive, 8), 1),
        so every ray still alive at the cap has received the same frame-wide C_eff = sum of n_step samples, in [max_steps, max_steps + 7]
        (the deployed max_steps = 16 binds on most foreground rays).
          cap = "reference" (default): fused mode reproduces exactly that -- phase 1 to max_steps, a device-side replay of the schedule from
            a histogram over the rays, phase 2 to C_eff -- and equals the loop under the reference schedule (1, 8) bit for bit: pixels, depth,
            sums and, with count_samples, per-ray marched counts.  A tile of a larger frame passes the frame's ray count and a histogram
            exchange (dist.ShardedFrame.configure does both) and then equals the unsharded frame.
          cap = "per_ray": a ray alive at the cap stops at ceil(max_steps / S) * S samples, S = steps_per_pass -- the loop under the
            schedule (budget_factor, n_step_cap) = (S, S); one launch less, no exchange between ranks, NOT the reference's pixels on rays
            that reach the cap.
        Every ray that leaves the box or falls under T_thresh before max_steps -- all rays of the 512^2 / 192-step headline frame -- is
        the same under both."""
        if mode not in ("loop", "fused"):
            raise ValueError("mode must be 'loop' or 'fused'")
        if cap not in ("reference", "per_ray"):
            raise ValueError("cap must be 'reference' or 'per_ray'")
        self.mode = mode
        self.cap = cap
        # fused mode, cap "reference", when this renderer draws a TILE of a frame: the frame's ray count (the N of renderer.py:513) and a
        # callable that sums the [max_steps + 1] int32 histogram over the ranks in place (dist.ShardedFrame.configure sets both)
        self.frame_rays_total = None
        self.cap_exchange = None
        self.steps_per_pass = 0    # fused mode: samples per ray and pass (0 = auto by ray count; 1, 2, 4, 8, 16 = the schedule n_step it equals)
        import math
        self.head = head
        self.bound = float(bound)
        self.cascade = cascade if cascade is not None else 1 + math.ceil(math.log2(bound))  # renderer.py:93
        self.grid_size = grid_size
        self.bitfield = density_bitfield.contiguous()
        dev = self.bitfield.device
        if aabb is None:  # renderer.py:110 -- y extent halved
            aabb = torch.tensor([-bound, -bound / 2, -bound, bound, bound / 2, bound], dtype=torch.float32, device=dev)
        self.aabb = aabb.to(dev, torch.float32).contiguous()
        self._aabb_diag = None      # (tensor, version, diagonal): see _cap_can_bind
        self.min_near = float(min_near)
        if density_scale != 1:
            raise NotImplementedError("density_scale != 1 (the reference hard-codes 1, renderer.py:95)")
        # iteration schedule n_step = max(min(budget_factor * N // n_alive, n_step_cap), 1); (1, 8) is the reference's
        # (renderer.py:513).  A larger budget trades HBM (288 GB here) for fewer, fatter iterations; pixels are unchanged.
        self.budget_factor, self.n_step_cap = int(budget_factor), int(n_step_cap)
        self._buf = None
        self._head_events = None  # set to a list: Python-driven loop with torch events around every head launch (debug)
        self._timing = None       # lz_timing handle: native loop records a HIP event pair around every head launch
        self.chunk = 8       # iterations enqueued per C call
        self.lookahead = 2   # chunks the host keeps queued ahead of the one whose result it inspects (ring holds lookahead + 2)
        # fused mode: confine every ray's march to the bounds of the occupied cells (lz_occupied_bounds, recomputed when the bitfield
        # tensor changes): same samples, bit for bit, without a cell test per empty cell in front of / behind the object
        self.clip_to_occupancy = True
        self.occupancy_margin = 8      # cells; see occupied_bounds()
        self._occ = None

    def occupied_bounds(self):
        """device tensor [6]: world-space box of the occupied cells of `self.bitfield`, dilated by `occupancy_margin` cells of each level's own size and by
        at least four of the march's longest steps (dt_max = sqrt(3) cells of the outermost level), so that the march walks its last steps
        in front of the first occupied cell with its ordinary cell tests.  Cached per (tensor object, version): an in-place update of the bitfield through torch or through this package
        (occupancy.update_density_grid, raymarching.packbits into a supplied bitfield: both bump the version) recomputes it; code that writes
        the bitfield through its raw pointer by other means calls invalidate_occupancy()."""
        bf = self.bitfield
        # the entry holds the tensor itself: identity (not the address, which the caching allocator hands to the next bitfield of the same
        # size) plus the version counter.  Inference tensors (created under torch.inference_mode, e.g. a checkpoint loaded inside a serving
        # loop) track no version: they are rescanned every frame (two small launches)
        version = None if bf.is_inference() else bf._version
        hit = (self._occ is not None and version is not None and self._occ[0] is bf and self._occ[1] == (version, self.occupancy_margin, self.cascade,
                                                                                                      self.grid_size, self.bound))
        if not hit:
            box = torch.empty(6, dtype=torch.float32, device=bf.device)
            ws = torch.empty(48, dtype=torch.int32, device=bf.device)
            call("lz_occupied_bounds", ptr(bf), int(self.cascade), int(self.grid_size), self.bound, int(self.occupancy_margin), ptr(ws), ptr(box), stream())
            self._occ = (bf, (version, self.occupancy_margin, self.cascade, self.grid_size, self.bound), box, ws)
        return self._occ[2]

    def _cap_can_bind(self, dt_gamma, max_steps):
        """False when no ray can collect max_steps samples, so the cap -- and with it the iteration schedule -- cannot touch any pixel: a
        ray's samples lie inside the aabb, at most its diagonal D apart, and consecutive samples are at least dt_min = min(dt_max, 2 sqrt(3) /
        max_steps) apart (raymarching.cu:866-867, 907), so a ray holds at most D / dt_min + 1 of them.  The reference's own aabb (y extent
        halved, renderer.py:110: D = 3 bound < 2 sqrt(3) bound) never reaches the cap once dt_min = 2 sqrt(3) / max_steps, e.g. the 192-step
        headline frame; its deployed max_steps = 16 (dt_min = dt_max) does.  The diagonal is read back once per aabb tensor."""
        a = self.aabb
        ver = None if a.is_inference() else a._version
        if self._aabb_diag is None or self._aabb_diag[0] is not a or self._aabb_diag[1] != ver or ver is None:
            lo_hi = a.detach().cpu().double()
            self._aabb_diag = (a, ver, float(((lo_hi[3:] - lo_hi[:3]) ** 2).sum().sqrt()))
        import math
        dt_max = 2 * math.sqrt(3) * (1 << (int(self.cascade) - 1)) / int(self.grid_size)
        dt_min = min(dt_max, 2 * math.sqrt(3) / max(int(max_steps), 1))
        return self._aabb_diag[2] / dt_min + 2 >= int(max_steps)

    def invalidate_occupancy(self):
        """forget the cached bounds of the occupied cells (the next fused frame rescans the bitfield)"""
        self._occ = None

    def timing_start(self, n_pairs):
        """bracket every head launch of the following render() calls with HIP events (bench.py's roofline leg)"""
        h = C.c_void_p()
        call("lz_timing_create", int(n_pairs), C.byref(h))
        self._timing = h

    def timing_stop(self):
        """-> list of head-launch durations in ms (synchronises the device)"""
        torch.cuda.synchronize()
        h, self._timing = self._timing, None
        cap = 1 << 20
        buf = (C.c_float * cap)()
        n = C.c_uint32()
        call("lz_timing_elapsed_ms", h, buf, cap, C.byref(n))
        out = list(buf[: n.value])
        call("lz_timing_destroy", h)
        return out

    def _buffers(self, N, device):
        rows = max(N * self.budget_factor, N)
        if self._buf is None or self._buf.N != N or self._buf.rows != rows:
            self._buf = _Buffers(N, rows, device)
        return self._buf

    def _iteration(self, b, cur, N, enc_a, ind_code, eye, dt_gamma, max_steps, T_thresh, count_samples):
        st, ws = ptr(b.state), ptr(b.workspace)
        nxt = 1 - cur
        # compaction of list[cur] -> list[nxt], fused with the march of the survivors
        call("lz_loop_march", st, N, N * self.budget_factor, self.n_step_cap, ptr(b.rays_alive[cur]), ptr(b.rays_alive[nxt]), ws, ptr(b.rays_t), ptr(self._rays_o),
             ptr(self._rays_d), self.bound, float(dt_gamma), int(max_steps), int(self.cascade), int(self.grid_size), ptr(self.bitfield),
             ptr(b.nears), ptr(b.fars), ptr(b.xyzs), ptr(b.dirs), ptr(b.deltas), ptr(b.ray_counts) if count_samples else None, stream())
        ev = self._head_events
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.head.forward(b.xyzs, b.dirs, enc_a, ind_code, eye, testing=True, count_ptr=b.state.data_ptr() + _N_SAMPLES_OFF,
                          out=(b.sigmas, b.rgbs, b.amb_aud, b.amb_eye, b.unc))
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        call("lz_loop_composite", st, N, float(T_thresh), ptr(b.rays_alive[nxt]), ptr(b.rays_t), ptr(b.sigmas), ptr(b.rgbs),
             ptr(b.deltas), ptr(b.amb_aud), ptr(b.amb_eye), ptr(b.unc), ptr(b.weights_sum), ptr(b.depth), ptr(b.image),
             ptr(b.amb_aud_sum), ptr(b.amb_eye_sum), ptr(b.unc_sum), ws, stream())

    def _frame(self, b, N, enc_a, ind_code, eye, dt_gamma, max_steps, T_thresh, count_samples):
        """lz_frame for lz_loop_run; the small conditioning tensors are kept alive on self until the next frame"""
        h = self.head
        enc_a = enc_a.reshape(-1).float().contiguous()
        ind_code = None if ind_code is None else ind_code.reshape(-1).float().contiguous()
        eye = None if eye is None else eye.reshape(-1).float().contiguous()
        self._cond = (enc_a, ind_code, eye)
        f = _lib.Frame()
        f.head = h._params(enc_a, ind_code, eye, True)
        p = lambda t: t.data_ptr()
        f.state, f.workspace = p(b.state), p(b.workspace)
        f.rays_alive[0], f.rays_alive[1] = p(b.rays_alive[0]), p(b.rays_alive[1])
        f.rays_t, f.rays_o, f.rays_d, f.nears, f.fars, f.grid = p(b.rays_t), p(self._rays_o), p(self._rays_d), p(b.nears), p(b.fars), p(self.bitfield)
        f.xyzs, f.dirs, f.deltas = p(b.xyzs), p(b.dirs), p(b.deltas)
        f.sigmas, f.rgbs, f.amb_aud, f.amb_eye, f.unc = p(b.sigmas), p(b.rgbs), p(b.amb_aud), p(b.amb_eye), p(b.unc)
        f.weights_sum, f.depth, f.image = p(b.weights_sum), p(b.depth), p(b.image)
        f.amb_aud_sum, f.amb_eye_sum, f.unc_sum = p(b.amb_aud_sum), p(b.amb_eye_sum), p(b.unc_sum)
        f.ray_counts = p(b.ray_counts) if count_samples else None
        f.N, f.max_steps, f.C, f.H = N, int(max_steps), int(self.cascade), int(self.grid_size)
        f.bound, f.dt_gamma, f.T_thresh = self.bound, float(dt_gamma), float(T_thresh)
        f.sample_budget, f.n_step_cap = N * self.budget_factor, self.n_step_cap
        return f

    @torch.no_grad()
    def render(self, rays_o, rays_d, enc_a, ind_code=None, eye=None, dt_gamma=1.0 / 256, max_steps=16, T_thresh=1e-4,
               bg_color=1.0, count_samples=False, sync_free=True, rgb24=False, perturb=False, noises=None):
        """rays_o, rays_d: [N,3] (or [1,N,3]) f32 cuda.  Returns dict(image [N,3] blended+clamped, weights_sum, depth,
        amb_aud_sum, amb_eye_sum, uncertainty_sum, state (device int32[8]), ray_counts if requested).
        perturb: the reference's inference loop passes `perturb` to march_rays on its FIRST iteration only (renderer.py:344,521), where
        every ray's start moves by clamp(near * dt_gamma, dt_min, dt_max) * U[0,1) (raymarching.cu:873).  `noises` [N] supplies the
        draws (tests; default torch.rand like raymarching.py:298).
        sync_free: the host never waits for the newest work, only for the chunk `lookahead` chunks back (the GPU
        queue never drains); the returned tensors are ready in stream order."""
        rays_o = rays_o.reshape(-1, 3).float().contiguous()
        rays_d = rays_d.reshape(-1, 3).float().contiguous()
        N = rays_o.shape[0]
        if perturb and noises is None:
            noises = torch.rand(N, dtype=torch.float32, device=rays_o.device)
        if noises is not None:
            noises = noises.reshape(-1).to(rays_o.device, torch.float32).contiguous()
            if noises.numel() != N:
                raise RuntimeError("noises must hold one value per ray")
        if N > MAX_RAYS_PER_PASS and self.mode != "fused":
            # rays are independent: larger batches are rendered in passes of <= 2^20 rays (the device loop scans at most 4096
            # workgroup counts per iteration) and concatenated; pixels do not depend on the split
            outs = []
            for lo in range(0, N, MAX_RAYS_PER_PASS):
                hi = min(lo + MAX_RAYS_PER_PASS, N)
                bg = bg_color[lo:hi] if torch.is_tensor(bg_color) and bg_color.dim() > 1 and bg_color.shape[0] == N else bg_color
                o = self.render(rays_o[lo:hi], rays_d[lo:hi], enc_a, ind_code, eye, dt_gamma, max_steps, T_thresh, bg, count_samples,
                                sync_free, rgb24, noises=None if noises is None else noises[lo:hi])
                outs.append({k: v.clone() for k, v in o.items()})
            res = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0] if k not in ("state",)}
            st = torch.stack([o["state"] for o in outs])
            res["state"] = st[-1].clone()
            res["state"][5] = st[:, 5].sum()      # marched samples of the whole batch
            res["state"][72] = st[:, 72].sum()
            res["state"][6] = st[:, 6].max()
            return res
        if self.mode == "fused":
            return self._render_fused(rays_o, rays_d, enc_a, ind_code, eye, dt_gamma, max_steps, T_thresh, bg_color, count_samples, rgb24, noises)
        if N == 0:
            return _empty_result(rays_o.device, count_samples, rgb24)
        b = self._buffers(N, rays_o.device)
        self._rays_o, self._rays_d = rays_o, rays_d
        call("lz_near_far_from_aabb", ptr(rays_o), ptr(rays_d), ptr(self.aabb), N, self.min_near, ptr(b.nears), ptr(b.fars), stream())
        if count_samples:
            b.ray_counts.zero_()
        starts = b.nears
        if noises is not None:   # the loop's first march starts every ray at its perturbed t (the kernel copies `starts` into rays_t)
            starts = torch.empty_like(b.nears)
            call("lz_perturb_starts", ptr(b.nears), ptr(noises), float(dt_gamma), int(max_steps), int(self.cascade), int(self.grid_size), N, ptr(starts), stream())
        call("lz_loop_begin", N, int(max_steps), N * self.budget_factor, self.n_step_cap, ptr(starts), ptr(b.rays_alive[0]), ptr(b.rays_t), ptr(b.weights_sum), ptr(b.depth),
             ptr(b.image), ptr(b.amb_aud_sum), ptr(b.amb_eye_sum), ptr(b.unc_sum), ptr(b.state), ptr(b.workspace), stream())
        cur, it = 0, 0
        pending_q = []
        native = self._head_events is None   # the Python-driven loop is kept for debugging with torch events
        if native:
            frame = self._frame(b, N, enc_a, ind_code, eye, dt_gamma, max_steps, T_thresh, count_samples)
        limit = int(max_steps) + 1   # n_step >= 1: max_steps iterations always suffice (renderer.py:503,546); the state of
        while it < limit:            # iteration k is committed by iteration k + 1's launches, hence one more
            n = min(self.chunk, limit - it)
            if native:   # one C call enqueues n x (march, head, composite, advance)
                call("lz_loop_run", C.byref(frame), cur, n, self._timing, stream())
                cur = (cur + n) & 1
            else:
                for _ in range(n):
                    self._iteration(b, cur, N, enc_a, ind_code, eye, dt_gamma, max_steps, T_thresh, count_samples)
                    cur = 1 - cur
            it += n
            # After every chunk the loop state is copied to a pinned slot (ring of `lookahead + 1`) behind an event.
            # Before enqueuing more, the host looks at the chunk `lookahead` chunks back: the GPU always has that many
            # chunks queued (no bubble), and at most that many no-op chunks are enqueued after the frame has finished.
            k = len(pending_q)
            slot = b.state_ring[k % len(b.state_ring)]
            slot.copy_(b.state[:8], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending_q.append((ev, slot))
            look = 0 if not sync_free else self.lookahead
            if k >= look:
                ev_old, slot_old = pending_q[k - look]
                ev_old.synchronize()
                if int(slot_old[3]) == 1:
                    break
        bg = None
        bg_scalar = 1.0
        if torch.is_tensor(bg_color):
            bg = bg_color.to(rays_o.device, torch.float32).expand(N, 3).contiguous()
        else:
            bg_scalar = float(bg_color)
        if rgb24:   # + the video pipe's hand-off format, quantised on device (TrainerUtil.py:550-555)
            if getattr(b, "out_rgb24", None) is None:
                b.out_rgb24 = torch.empty(N, 3, dtype=torch.uint8, device=rays_o.device)
            call("lz_final_blend_rgb24", ptr(b.image), ptr(b.weights_sum), ptr(bg), bg_scalar, N, ptr(b.out), ptr(b.out_rgb24), stream())
        else:
            call("lz_final_blend", ptr(b.image), ptr(b.weights_sum), ptr(bg), bg_scalar, N, ptr(b.out), stream())
        res = dict(image=b.out, image_raw=b.image, weights_sum=b.weights_sum, depth=b.depth, amb_aud_sum=b.amb_aud_sum,
                   amb_eye_sum=b.amb_eye_sum, uncertainty_sum=b.unc_sum, state=b.state, nears=b.nears, fars=b.fars)
        if count_samples:
            res["ray_counts"] = b.ray_counts
        if rgb24:
            res["image_rgb24"] = b.out_rgb24
        return res

    # ---- the whole frame as one persistent kernel (csrc/lz_frame.hip) ----
    def _fused_buffers(self, N, device):
        fb = getattr(self, "_fbuf", None)
        if fb is None or fb["N"] != N or fb["device"] != device:
            f = dict(dtype=torch.float32, device=device)
            fb = dict(N=N, device=device, nears=torch.empty(N, **f), fars=torch.empty(N, **f), rays_t=torch.empty(N, **f),
                      order=torch.empty(N, dtype=torch.int32, device=device), state=torch.zeros(1024, dtype=torch.int32, device=device),
                      keys=torch.empty(N, dtype=torch.uint8, device=device), weights_sum=torch.empty(N, **f), depth=torch.empty(N, **f),
                      image=torch.empty(N, 3, **f), amb_aud_sum=torch.empty(N, **f), amb_eye_sum=torch.empty(N, **f), unc_sum=torch.empty(N, **f),
                      out=torch.empty(N, 3, **f), out_rgb24=None, ray_counts=None, t_end=torch.empty(N, **f),
                      ray_last=torch.empty(N, dtype=torch.int32, device=device), cap_ws=None)
            self._fbuf = fb
        return fb

    def _render_fused(self, rays_o, rays_d, enc_a, ind_code, eye, dt_gamma, max_steps, T_thresh, bg_color, count_samples, rgb24, noises=None):
        ctx = self.fused_begin(rays_o, rays_d, enc_a, ind_code, eye, dt_gamma, max_steps, T_thresh, bg_color, count_samples, rgb24, noises)
        if ctx["deferred"] and self.cap_exchange is not None:
            self.cap_exchange(ctx["hist"])
        return self.fused_finish(ctx)

    def fused_begin(self, rays_o, rays_d, enc_a, ind_code=None, eye=None, dt_gamma=1.0 / 256, max_steps=16, T_thresh=1e-4, bg_color=1.0,
                    count_samples=False, rgb24=False, noises=None):
        """Fused mode in two calls, for callers that render tiles of ONE frame under cap = "reference": fused_begin enqueues phase 1 and the
        histogram of the rays' last surviving chunk boundary -- ctx["hist"], int32 [max_steps + 1] on the device; the caller sums it over all
        tiles of the frame, in place -- and fused_finish(ctx) enqueues the schedule replay, phase 2 and returns the result dict.  With
        frame_rays_total unset (a whole frame) and for cap = "per_ray" everything happens in fused_begin."""
        rays_o = rays_o.reshape(-1, 3).float().contiguous()
        rays_d = rays_d.reshape(-1, 3).float().contiguous()
        N, dev = rays_o.shape[0], rays_o.device
        b = self._fused_buffers(N, dev)
        h = self.head
        enc_a = enc_a.reshape(-1).float().contiguous()
        ind_code = None if ind_code is None else ind_code.reshape(-1).float().contiguous()
        eye = None if eye is None else eye.reshape(-1).float().contiguous()
        bg = None
        if torch.is_tensor(bg_color):
            bg = bg_color.to(dev, torch.float32).expand(N, 3).contiguous()
        if rgb24 and b["out_rgb24"] is None:
            b["out_rgb24"] = torch.empty(N, 3, dtype=torch.uint8, device=dev)
        if count_samples and b["ray_counts"] is None:
            b["ray_counts"] = torch.zeros(N, dtype=torch.int32, device=dev)
        f = _lib.FrameFused()
        f.head = h._params(enc_a, ind_code, eye, True)
        p = lambda t: None if t is None else t.data_ptr()
        f.rays_o, f.rays_d, f.grid, f.aabb = p(rays_o), p(rays_d), p(self.bitfield), p(self.aabb)
        for k in ("nears", "fars", "rays_t", "order", "state", "keys", "weights_sum", "depth", "image", "amb_aud_sum", "amb_eye_sum", "unc_sum", "out"):
            setattr(f, k, p(b[k]))
        f.bg, f.out_rgb24 = p(bg), p(b["out_rgb24"]) if rgb24 else None
        f.ray_counts = p(b["ray_counts"]) if count_samples else None
        f.bg_scalar = 1.0 if bg is not None else float(bg_color)
        f.bound, f.dt_gamma, f.T_thresh, f.min_near = self.bound, float(dt_gamma), float(T_thresh), self.min_near
        f.N, f.max_steps, f.C, f.H = N, int(max_steps), int(self.cascade), int(self.grid_size)
        f.steps_per_pass = int(self.steps_per_pass)
        f.noises = None if noises is None else noises.data_ptr()
        if self.clip_to_occupancy:
            f.occupied_aabb, f.t_end = p(self.occupied_bounds()), p(b["t_end"])
        deferred = False
        # the reference's cap needs the schedule replay only if some ray can reach max_steps at all (or marched counts are asked for: a
        # T_thresh-cut ray's count depends on the chunk it was cut in); otherwise the plain launch renders the same pixels
        # (ranks rendering tiles of one frame share aabb and max_steps, so they take the same branch: no histogram exchange either)
        if self.cap == "reference" and (count_samples or self._cap_can_bind(dt_gamma, max_steps)):
            need = 2 * int(max_steps) + 24            # LZ_FRAME_CAP_WS_INTS
            if b["cap_ws"] is None or b["cap_ws"].numel() < need:
                b["cap_ws"] = torch.zeros(need, dtype=torch.int32, device=dev)
            f.cap_mode, f.ray_last, f.cap_ws = 1, p(b["ray_last"]), p(b["cap_ws"])
            if self.frame_rays_total is None and _SPLIT_CAP:      # diagnostic: the two cap kernels as separate launches (tools/frame_trace.sh)
                f.N_total, f.defer_finish = N, 1
                deferred = True
            if self.frame_rays_total is not None:
                if int(self.frame_rays_total) < N:
                    raise ValueError("frame_rays_total is the ray count of the whole frame (>= the rays of this call)")
                f.N_total, f.defer_finish = int(self.frame_rays_total), 1
                deferred = True
        call("lz_frame_render", C.byref(f), self._timing, stream())   # timing: one event pair around the persistent kernel
        keep = (enc_a, ind_code, eye, bg, rays_o, rays_d, noises)
        self._keep = keep
        return dict(f=f, b=b, keep=keep, deferred=deferred, count_samples=count_samples, rgb24=rgb24,
                    hist=b["cap_ws"][: int(max_steps) + 1] if deferred else None)

    def fused_finish(self, ctx):
        b = ctx["b"]
        if ctx["deferred"]:
            call("lz_frame_finish", C.byref(ctx["f"]), stream())
        res = dict(image=b["out"], image_raw=b["image"], weights_sum=b["weights_sum"], depth=b["depth"], amb_aud_sum=b["amb_aud_sum"],
                   amb_eye_sum=b["amb_eye_sum"], uncertainty_sum=b["unc_sum"], state=b["state"], nears=b["nears"], fars=b["fars"])
        if ctx["count_samples"]:
            res["ray_counts"] = b["ray_counts"]
        if ctx["rgb24"]:
            res["image_rgb24"] = b["out_rgb24"]
        return res


class NetworkRenderer(TriplaneRenderer):
    """The device-resident inference loop around ANY per-sample network (BASELINE cfg2: a generic hash-grid NeRF on the operator API):
    run_cuda_for_inference's iteration (renderer.py:503-548) as lz_loop_march -> net -> lz_loop_composite with the loop state in device
    memory, like TriplaneRenderer(mode="loop") -- but the network is a Python callable over torch tensors,

        net(xyzs [rows, 3], dirs [rows, 3]) -> (sigma [rows], rgb [rows, 3])

    evaluated every iteration on the WHOLE row budget (rows = budget_factor * N, static shapes): rows behind the iteration's n_alive *
    n_step samples hold older samples whose results nobody reads.  That costs a few useless rows (the budget is what the schedule
    fills while most rays are alive) and buys an iteration without a host round trip -- the reference syncs on `rays_alive[rays_alive >=
    0]` every iteration -- and with static shapes, so that a pair of iterations (the alive list ping-pongs) is captured ONCE as a hipGraph
    (torch.cuda.CUDAGraph over torch's kernels and this library's launches on the capturing stream) and replayed: the ~25 launches of an
    iteration cost one graph launch.  Pixels and per-ray sample counts equal the reference loop's on the same operators (rays are
    independent, compositing resumes exactly; tests/test_gpu_cfg2_render.py)."""

    def __init__(self, net, density_bitfield, bound=1.0, cascade=None, grid_size=128, aabb=None, min_near=0.05, budget_factor=4, n_step_cap=4,
                 graph=True):
        super().__init__(None, density_bitfield, bound=bound, cascade=cascade, grid_size=grid_size, aabb=aabb, min_near=min_near,
                         budget_factor=budget_factor, n_step_cap=n_step_cap, mode="loop")
        self.net = net
        self.use_graph = bool(graph)
        self._graph = None        # (key, CUDAGraph) of two iterations

    def _pair(self, b, N, dt_gamma, max_steps, T_thresh, count_samples):
        st, ws = ptr(b.state), ptr(b.workspace)
        for cur in (0, 1):
            nxt = 1 - cur
            call("lz_loop_march", st, N, N * self.budget_factor, self.n_step_cap, ptr(b.rays_alive[cur]), ptr(b.rays_alive[nxt]), ws, ptr(b.rays_t),
                 ptr(self._rays_o), ptr(self._rays_d), self.bound, float(dt_gamma), int(max_steps), int(self.cascade), int(self.grid_size),
                 ptr(self.bitfield), ptr(b.nears), ptr(b.fars), ptr(b.xyzs), ptr(b.dirs), ptr(b.deltas), ptr(b.ray_counts) if count_samples else None,
                 stream())
            sigma, rgb = self.net(b.xyzs, b.dirs)
            b.sigmas.copy_(sigma.reshape(-1))
            b.rgbs.copy_(rgb.reshape(-1, 3))
            call("lz_loop_composite", st, N, float(T_thresh), ptr(b.rays_alive[nxt]), ptr(b.rays_t), ptr(b.sigmas), ptr(b.rgbs), ptr(b.deltas),
                 ptr(b.amb_aud), ptr(b.amb_eye), ptr(b.unc), ptr(b.weights_sum), ptr(b.depth), ptr(b.image), ptr(b.amb_aud_sum), ptr(b.amb_eye_sum),
                 ptr(b.unc_sum), ws, stream())

    @torch.no_grad()
    def render(self, rays_o, rays_d, dt_gamma=1.0 / 256, max_steps=128, T_thresh=1e-4, bg_color=1.0, count_samples=False):
        """-> dict(image [N,3] blended + clamped, image_raw, weights_sum, depth, state, ray_counts if requested); ready in stream order"""
        rays_o = rays_o.reshape(-1, 3).float().contiguous()
        rays_d = rays_d.reshape(-1, 3).float().contiguous()
        N, dev = rays_o.shape[0], rays_o.device
        if N > MAX_RAYS_PER_PASS:
            raise RuntimeError("NetworkRenderer renders at most %d rays per call" % MAX_RAYS_PER_PASS)
        if N == 0:
            return _empty_result(dev, count_samples, ambient=False)
        prev = self._buf
        b = self._buffers(N, dev)
        fresh = b is not prev      # (also when the row budget changed: budget_factor)
        if fresh:   # rows no march has written yet are evaluated too: give the network finite positions there
            b.xyzs.zero_(); b.dirs.zero_(); b.deltas.zero_()
            b.amb_aud.zero_(); b.amb_eye.zero_(); b.unc.zero_()
            self._graph = None
        # the graph holds the addresses of the ray tensors: keep them in buffers of our own
        if getattr(b, "ro", None) is None:
            b.ro, b.rd = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
        b.ro.copy_(rays_o); b.rd.copy_(rays_d)
        self._rays_o, self._rays_d = b.ro, b.rd
        call("lz_near_far_from_aabb", ptr(b.ro), ptr(b.rd), ptr(self.aabb), N, self.min_near, ptr(b.nears), ptr(b.fars), stream())
        if count_samples:
            b.ray_counts.zero_()
        call("lz_loop_begin", N, int(max_steps), N * self.budget_factor, self.n_step_cap, ptr(b.nears), ptr(b.rays_alive[0]), ptr(b.rays_t), ptr(b.weights_sum),
             ptr(b.depth), ptr(b.image), ptr(b.amb_aud_sum), ptr(b.amb_eye_sum), ptr(b.unc_sum), ptr(b.state), ptr(b.workspace), stream())
        # everything the captured launches bake in: scalars, the addresses of the bitfield / aabb / buffers, the schedule
        key = (N, float(dt_gamma), int(max_steps), float(T_thresh), bool(count_samples), self.bitfield.data_ptr(), self.aabb.data_ptr(), self.bound,
               int(self.cascade), int(self.grid_size), self.n_step_cap, self.budget_factor, self.min_near, id(b))
        if self.use_graph and (self._graph is None or self._graph[0] != key):
            # capture two iterations.  The capture itself executes nothing, and torch wants the ops warmed up on a side stream first:
            # run one pair for real there (it is simply the frame's first pair), then capture
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._pair(b, N, dt_gamma, max_steps, T_thresh, count_samples)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._pair(b, N, dt_gamma, max_steps, T_thresh, count_samples)
            self._graph = (key, g)
            done_pairs = 1
        else:
            done_pairs = 0
        limit = (int(max_steps) + 2) // 2 + 1          # pairs: n_step >= 1, plus the iteration that commits the last state
        pending = []
        it = done_pairs
        while it < limit:
            n = min(max(self.chunk // 2, 1), limit - it)
            for _ in range(n):
                if self.use_graph:
                    self._graph[1].replay()
                else:
                    self._pair(b, N, dt_gamma, max_steps, T_thresh, count_samples)
            it += n
            k = len(pending)
            slot = b.state_ring[k % len(b.state_ring)]
            slot.copy_(b.state[:8], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending.append((ev, slot))
            if k >= self.lookahead:
                ev_old, slot_old = pending[k - self.lookahead]
                ev_old.synchronize()
                if int(slot_old[3]) == 1:
                    break
        bg = bg_color.to(dev, torch.float32).expand(N, 3).contiguous() if torch.is_tensor(bg_color) else None
        call("lz_final_blend", ptr(b.image), ptr(b.weights_sum), ptr(bg), 1.0 if bg is not None else float(bg_color), N, ptr(b.out), stream())
        self._keep = bg
        res = dict(image=b.out, image_raw=b.image, weights_sum=b.weights_sum, depth=b.depth, state=b.state)
        if count_samples:
            res["ray_counts"] = b.ray_counts
        return res
