"""Multi-GPU: camera rays are independent, so a batch of rays is sharded contiguously across ranks (one process per
GPU), every rank renders its shard with a full replica of the model (tables 1.96 MB + MLPs 93 KB + bitfield 256 KB),
and the rendered tiles are all-gathered with RCCL over xGMI.  The reference has no reachable distributed code
(dead DDP wrapper only); this is new.

Payloads are tiny (512x512 frame = 3 MB of f32 RGB, 768 KB as RGB24), i.e. latency-bound on the 7 point-to-point
xGMI links: ONE all_gather per frame batch, uint8 when the consumer is the video pipe (TrainerUtil.py:550-555
quantises to uint8 before hand-off anyway), never per-iteration collectives.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:   # LZ_DIST_BACKEND=gloo: rehearse the multi-process path on a box with fewer GPUs than ranks
            backend = os.environ.get("LZ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" is RCCL on ROCm
        if torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", rank))
            torch.cuda.set_device(local % torch.cuda.device_count() if backend != "nccl" else local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_bounds(n_rays, rank, world):
    """contiguous, balanced split of n_rays: the first (n_rays % world) ranks get one extra ray"""
    base, rem = divmod(n_rays, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(rays_o, rays_d, rank, world):
    lo, hi = shard_bounds(rays_o.shape[0], rank, world)
    return rays_o[lo:hi].contiguous(), rays_d[lo:hi].contiguous()


STRIPE_ROWS = 8   # interleaved tiling: stripes of 8 image rows dealt round-robin


def tile_rows(H, rank, world, tiles="contiguous"):
    """image rows of rank `rank`'s tile(s): "contiguous" = one block of rows (shard_bounds over rows), "interleaved" = stripes of
    STRIPE_ROWS rows dealt round-robin -- the face in the middle of the frame costs more samples per ray than the margins, and
    stripes spread it over all ranks.  Returns a python list of row indices (ascending)."""
    if tiles == "contiguous":
        lo, hi = shard_bounds(H, rank, world)
        return list(range(lo, hi))
    if tiles != "interleaved":
        raise ValueError("tiles must be 'contiguous' or 'interleaved'")
    return [r for r in range(H) if (r // STRIPE_ROWS) % world == rank]


def tile_pixels(H, W, rank, world, tiles="contiguous", device="cpu"):
    """int64 [n_local] flat pixel indices (row * W + col) of this rank's tile(s), ascending; None for a single rank (whole frame)"""
    if world == 1:
        return None
    rows = torch.tensor(tile_rows(H, rank, world, tiles), dtype=torch.int64, device=device)
    return (rows[:, None] * W + torch.arange(W, dtype=torch.int64, device=device)[None, :]).reshape(-1)


def frame_permutation(H, W, world, tiles="contiguous", device="cpu"):
    """pixel index of every row of the rank-major gathered buffer (rank 0's tile, rank 1's, ...): frame[perm] = gathered.
    With equal tiles it is the concatenation of every rank's tile_pixels; identity for contiguous tiles."""
    return torch.cat([tile_pixels(H, W, r, world, tiles, device) for r in range(world)]) if world > 1 else torch.arange(H * W, device=device)


def assemble_frame(gathered, H, W, world, tiles="contiguous"):
    """rank-major gathered tiles [H*W, C] -> the frame in pixel order [H*W, C] (a no-op for contiguous tiles)"""
    if world == 1 or tiles == "contiguous":
        return gathered
    out = torch.empty_like(gathered)
    out[frame_permutation(H, W, world, tiles, gathered.device)] = gathered
    return out


class TileGatherer:
    """ONE all-gather per frame, overlapped with the next frame's march: the rendered tile is copied to one of two
    staging buffers on the render stream, and the collective runs on a side stream behind an event, so frame k + 1 renders while
    frame k's tiles cross xGMI.  `gather(tile)` returns the gathered buffer of THIS frame; it is complete once `wait()` (or a
    device synchronisation) has run, and stays valid until the second-next call.  CPU tensors (gloo tests) take the same code
    path without streams."""

    def __init__(self, n_local, channels, dtype, device, sizes=None, group=None):
        self.group, self.sizes = group, sizes
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = torch.device(device).type == "cuda"
        self.stage = [torch.empty(n_local, channels, dtype=dtype, device=device) for _ in range(2)]
        self.out = [None, None]
        self.done = [None, None]
        self.k = 0
        self.comm = torch.cuda.Stream(device=device) if (self.cuda and self.world > 1) else None

    def gather(self, tile):
        if self.world == 1:
            return tile
        i = self.k & 1
        self.k += 1
        if self.comm is None:
            self.stage[i].copy_(tile)
            self.out[i] = gather_tiles(self.stage[i], None, self.group, self.sizes)
            return self.out[i]
        cur = torch.cuda.current_stream()
        if self.done[i] is not None:
            cur.wait_event(self.done[i])        # the collective that last read this staging buffer has finished
        self.stage[i].copy_(tile)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            self.out[i] = gather_tiles(self.stage[i], None, self.group, self.sizes)
            self.done[i] = torch.cuda.Event()
            self.done[i].record(self.comm)
        return self.out[i]

    def wait(self):
        """make the current stream wait for every collective issued so far"""
        if self.comm is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.comm)
            for o in self.out:       # the gathered buffers were allocated on the side stream: tell the allocator this stream reads them
                if o is not None:
                    o.record_stream(cur)


class PeerTileGatherer:
    """The same hand-off WITHOUT a collective (at ~390 KB per rank a ring all-gather is 7 dependent hops on point-to-point
    xGMI links; here every tile crosses exactly ONE hop).  Every rank owns two frame buffers [sum n_local, C] (double buffering, as
    TileGatherer) and exports them to its peers (CUDA / HIP IPC handles, exchanged once with all_gather_object); a frame then is

        for every peer p:  p.frame[k & 1][my_offset : my_offset + n_local] <- my tile        (W device-to-device copies, side stream)
                           p.flags[k & 1][me] <- k + 1                                       (a 4-byte copy BEHIND the data, same stream)
        wait on the device until my flags[k & 1][:] have all reached k + 1                   (lz_wait_flags: bounded spin, never hangs)

    `gather(tile)` returns this rank's own frame buffer of frame k, complete once `wait()` has run; valid until the second-next call.
    A buffer is reused two frames later: a peer may only overwrite it after this rank has finished reading frame k - 2, which the
    caller guarantees by calling `wait()` (and consuming the frame) before the second-next `gather` -- the flags of frame k - 1 that the
    peer waited for were written by this rank after that.
    An expired wait is FATAL, not silent: the device flag `timed_out` is copied to pinned host memory behind every wait kernel, and the
    next `gather()` (or an explicit `check()`, which synchronises) raises RuntimeError once it is set -- a peer skewed by more than the
    poll budget (first-frame warm-up, graph capture, a checkpoint load) would otherwise leave a torn frame in the returned buffer, and
    the buffer-reuse argument above would stop holding.  `max_polls` (~1 us each) sizes that budget per instance.

    CPU tensors (the gloo tests) take the same OFFSET / double-buffer / flag bookkeeping with the peer writes carried by a gloo
    all_gather: the indexing is what those tests pin; the transport needs the GPUs."""

    MAX_POLLS = 2_000_000   # ~2 s of polling before a wait gives up

    def __init__(self, n_local, channels, dtype, device, sizes=None, group=None, max_polls=None):
        self.group = group
        self.max_polls = int(max_polls) if max_polls is not None else self.MAX_POLLS
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.sizes = list(sizes) if sizes is not None else [n_local] * self.world
        if self.sizes[self.rank] != n_local:
            raise ValueError("PeerTileGatherer: sizes[rank] must equal n_local")
        self.offsets = [sum(self.sizes[:r]) for r in range(self.world)]
        self.total, self.n_local, self.channels = sum(self.sizes), n_local, channels
        self.cuda = torch.device(device).type == "cuda"
        self.k = 0
        self.frame = [torch.zeros(self.total, channels, dtype=dtype, device=device) for _ in range(2)]
        self.flags = [torch.zeros(max(self.world, 1), dtype=torch.int32, device=device) for _ in range(2)]
        self.timed_out = torch.zeros(1, dtype=torch.int32, device=device)
        self._timed_out_host = torch.zeros(1, dtype=torch.int32).pin_memory() if self.cuda else torch.zeros(1, dtype=torch.int32)
        self._checked = None       # event behind the newest host copy of `timed_out`
        self.comm = None
        self.peer_frame = self.peer_flags = None
        if self.world > 1 and self.cuda:
            self.comm = torch.cuda.Stream(device=device)
            self._tick = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(2)]
            self._open_peers(device)

    def _open_peers(self, device):
        """export my four buffers, import everyone's (same-process tensors for my own rank)"""
        mine = [t.untyped_storage()._share_cuda_() for t in self.frame + self.flags]
        meta = [(tuple(t.shape), t.dtype) for t in self.frame + self.flags]
        everyone = [None] * self.world
        dist.all_gather_object(everyone, (mine, meta), group=self.group)
        self.peer_frame, self.peer_flags = [], []
        for r, (handles, metas) in enumerate(everyone):
            if r == self.rank:
                ts = self.frame + self.flags
            else:
                ts = []
                for h, (shape, dt) in zip(handles, metas):
                    st = torch.UntypedStorage._new_shared_cuda(*h)
                    ts.append(torch.empty(0, dtype=dt, device=st.device).set_(st, 0, shape))
            self.peer_frame.append(ts[:2])
            self.peer_flags.append(ts[2:])
        dist.barrier(group=self.group)   # nobody proceeds (or frees) before every import has succeeded

    def check(self, synchronize=True):
        """raise if any wait so far has expired.  synchronize=False looks only at host copies that have already landed (what gather() does):
        the device word is sticky (lz_wait_flags only ever sets it), so whatever copy has reached the pinned host word is read regardless of
        the newest event -- an EARLIER frame's expired wait is seen even while the newest copy is still in flight"""
        if self._checked is not None:
            if synchronize:
                self._checked.synchronize()
            if int(self._timed_out_host[0]) != 0:
                raise RuntimeError("PeerTileGatherer: a wait for the peers' tiles expired after %d polls (rank %d, frame <= %d): the frame buffer is torn; "
                                   "raise max_polls or find the stalled peer" % (self.max_polls, self.rank, self.k))

    def gather(self, tile):
        if self.world == 1:
            return tile
        self.check(synchronize=False)      # the wait of an earlier frame expired: stop writing into peers' buffers
        i, want = self.k & 1, self.k + 1
        self.k += 1
        lo = self.offsets[self.rank]
        if self.comm is None:   # CPU: the same bookkeeping, the transport emulated by an all_gather into the offsets
            parts = [torch.empty(sz, self.channels, dtype=tile.dtype) for sz in self.sizes]
            if all(sz == self.sizes[0] for sz in self.sizes):
                dist.all_gather(parts, tile.contiguous(), group=self.group)
            else:
                mx = max(self.sizes)
                pad = torch.zeros(mx, self.channels, dtype=tile.dtype)
                pad[: tile.shape[0]] = tile
                padded = [torch.empty_like(pad) for _ in range(self.world)]
                dist.all_gather(padded, pad, group=self.group)
                parts = [p[:sz] for p, sz in zip(padded, self.sizes)]
            for r, part in enumerate(parts):          # what rank r's peer write leaves in MY buffer
                self.frame[i][self.offsets[r]: self.offsets[r] + self.sizes[r]] = part
                self.flags[i][r] = want
            return self.frame[i]
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)                              # the tile is rendered
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            self._tick[i].fill_(want)
            for d in range(self.world):                # start with my right-hand neighbour: the W ranks hit W different links at a time
                p = (self.rank + 1 + d) % self.world
                self.peer_frame[p][i][lo: lo + self.n_local].copy_(tile, non_blocking=True)
                self.peer_flags[p][i][self.rank: self.rank + 1].copy_(self._tick[i], non_blocking=True)
            from ._util import call, ptr
            call("lz_wait_flags", ptr(self.flags[i]), self.world, want, self.max_polls, ptr(self.timed_out), self.comm.cuda_stream)
            self._timed_out_host.copy_(self.timed_out, non_blocking=True)
            self._checked = torch.cuda.Event()
            self._checked.record(self.comm)
        tile.record_stream(self.comm)
        return self.frame[i]

    def wait(self, check=True):
        """order the current stream behind the hand-off of the newest frame.  check (default): also block the HOST until that frame's flag
        wait has reported and raise if it -- or any earlier one -- expired: the caller is about to consume the frame, a torn one must not
        pass silently (the error used to surface only in the NEXT gather(), never for the last frame).  check=False keeps the
        stream-only ordering for callers that poll check() themselves."""
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
            if check:
                self.check(synchronize=True)


class ShardedFrame:
    """BASELINE cfg4: ONE H x W frame whose rays are sharded over the ranks (row tiles, contiguous or interleaved stripes), every
    rank renders its tile with a full model replica, ONE all-gather per frame makes every rank hold the frame.

        sf = ShardedFrame(H, W, rank, world, tiles, device)
        rays_o, rays_d = sf.rays(pose, intrinsics)          # this rank's tile only (lz_get_rays reads the tile's pixel list)
        gathered = sf.gather(render(rays_o, rays_d))        # rank-major; overlapped with the next frame (TileGatherer)
        frame = sf.assemble(gathered)                       # pixel order (no-op for contiguous tiles)
    """

    def __init__(self, H, W, rank, world, tiles="contiguous", device="cuda", channels=3, dtype=torch.float32, group=None,
                 steps_per_pass=None, via="collective", cap="reference"):
        """cap (`configure(renderer)` applies it to a fused TriplaneRenderer): the reference stops every ray that is still alive at
        `max_steps` after the same FRAME-WIDE count C_eff = the sum of its loop's n_step = max(min(N // n_alive, 8), 1)
        (renderer.py:503-548) -- N and n_alive being those of the whole frame.
          "reference": the tiles reproduce that: every rank histograms its rays' last surviving chunk boundary, the max_steps + 1 int32
            words are summed over the ranks (ONE small all-reduce between the two phases of the frame kernel) and every rank replays the
            same schedule; the assembled frame equals the unsharded reference frame on every ray, whatever S each rank picked.
          "per_ray": no exchange; a ray at the cap stops at ceil(max_steps / S) * S samples, so only a pinned `steps_per_pass` makes
            tiles agree with each other (and none of them with the reference on those rays).
        steps_per_pass: pin the fused renderer's samples-per-ray-and-pass S (launch shape; otherwise chosen from the tile's ray count)."""
        if cap not in ("reference", "per_ray"):
            raise ValueError("cap must be 'reference' or 'per_ray'")
        self.cap, self.group = cap, group
        self.H, self.W, self.rank, self.world, self.tiles = H, W, rank, world, tiles
        self.steps_per_pass = steps_per_pass
        self.pixels = tile_pixels(H, W, rank, world, tiles, device)
        self.sizes = [len(tile_rows(H, r, world, tiles)) * W for r in range(world)]
        self.n_local = self.sizes[rank]
        if via not in ("collective", "peer"):
            raise ValueError("via must be 'collective' (one RCCL all-gather) or 'peer' (direct one-hop tile writes)")
        cls = PeerTileGatherer if via == "peer" else TileGatherer
        self.gatherer = cls(self.n_local, channels, dtype, device, self.sizes, group) if world > 1 else None

    def rays(self, pose, intrinsics):
        from .utils import frame_rays
        return frame_rays(pose, intrinsics, self.H, self.W, self.pixels)

    def sum_over_ranks(self, hist):
        """in-place sum of a small device tensor over the ranks of the frame (the cap histogram); a no-op for one rank"""
        if self.world > 1 and dist.is_initialized():
            dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=self.group)
        return hist

    def configure(self, renderer):
        """apply the cap rule and the pinned launch shape (if any) to a TriplaneRenderer; returns it"""
        if self.steps_per_pass is not None:
            renderer.steps_per_pass = int(self.steps_per_pass)
        renderer.cap = self.cap
        if self.cap == "reference" and self.world > 1:
            renderer.frame_rays_total = self.H * self.W
            renderer.cap_exchange = self.sum_over_ranks
        else:
            renderer.frame_rays_total = renderer.cap_exchange = None
        return renderer

    def gather(self, tile):
        return tile if self.gatherer is None else self.gatherer.gather(tile)

    def wait(self):
        if self.gatherer is not None:
            self.gatherer.wait()

    def assemble(self, gathered):
        return assemble_frame(gathered, self.H, self.W, self.world, self.tiles)


def cap_schedule_from_histogram(hist, n_total, max_steps):
    """Host mirror of lz_k_frame_schedule (csrc/lz_frame.hip), for tests and for reading a histogram: hist[l] = rays whose last surviving
    chunk boundary is l (bin max_steps: alive at the cap), summed over every tile of the frame.  Replays the reference's loop on the counts
    alone (renderer.py:503-548): n_alive at boundary B = rays with L >= B; n_step = max(min(N // n_alive, 8), 1); step += n_step while
    step < max_steps.  -> (C_eff, iterations, [chunk boundaries])."""
    h = [int(v) for v in hist]
    alive = [0] * (max_steps + 2)
    for b in range(max_steps, -1, -1):
        alive[b] = alive[b + 1] + (h[b] if b < len(h) else 0)
    B, bounds = 0, [0]
    while B < max_steps and alive[B] > 0:
        B += max(min(int(n_total) // alive[B], 8), 1)
        bounds.append(B)
    return B, len(bounds) - 1, bounds


def gather_tiles(tile, n_total=None, group=None, sizes=None):
    """all-gather per-rank tiles [n_local, C] into [sum n_local, C] on every rank, in rank order, with ONE collective.
    Equal tiles: a plain all_gather_into_tensor.  Ragged tiles: `sizes` = every rank's n_local (or `n_total`, meaning the
    contiguous `shard_bounds` split of n_total rays); tiles are padded to the largest, gathered once, and stripped."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tile
    world = dist.get_world_size(group)
    tile = tile.contiguous()
    if sizes is None and n_total is not None:
        sizes = [h - l for l, h in (shard_bounds(n_total, r, world) for r in range(world))]
    if sizes is None or all(sz == sizes[0] for sz in sizes):
        out = torch.empty((tile.shape[0] * world,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        dist.all_gather_into_tensor(out, tile, group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    pad[: tile.shape[0]] = tile
    out = torch.empty((mx * world,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + sz] for r, sz in enumerate(sizes)], 0)


def to_rgb24(image):
    """[N,3] f32 in [0,1] -> uint8, the hand-off format of the reference's video pipe (TrainerUtil.py:550-555)"""
    return (image * 255).to(torch.uint8)


class GradientBucket:
    """Data-parallel training over ray shards (training variant): every rank runs forward / backward on its shard of the
    sampled rays with a full replica, then the gradients are summed across ranks.  The whole model is small (3 x 163 584 table
    entries + ~24 k MLP weights + the audio nets, about 2.6 MB of f32), so instead of per-parameter or size-bucketed collectives all
    `.grad` tensors are VIEWS into one flat buffer and a step issues exactly ONE all-reduce over it -- latency-bound on xGMI, which is
    why it is one call.  Autograd accumulates into an existing `.grad` in place, so the views survive backward; clear gradients
    with `bucket.zero()` (an optimizer's `zero_grad(set_to_none=True)` would detach the views -- `attach()` re-creates them).

        bucket = GradientBucket(model.parameters())
        for batch in ...:
            bucket.zero(); loss(model, shard_of(batch)).backward(); bucket.all_reduce(); optimizer.step()
    """

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradientBucket: no parameter requires grad")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("GradientBucket: parameters must share one device and dtype")
        self.group = group
        n = sum(p.numel() for p in self.params)
        self._buf = torch.zeros(n + 1, dtype=dt, device=dev)   # last slot: this rank's weight (ray count), reduced in the same collective
        self.flat = self._buf[:n]
        self.attach()

    def attach(self):
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view(p.shape)
            off += n

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, average=True, weight=None):
        """One collective over the flat buffer; a no-op in a single process.

        Contract (what makes sharded training equal full-batch training):
        * `average=False`: plain sum -- for a SUM-reduced local loss (divide by the global count yourself);
        * `average=True, weight=None`: mean over ranks -- right only when every rank's loss is a mean over the SAME number of
          terms (equal ray shards and a per-ray loss);
        * `average=True, weight=w`: sum_r w_r g_r / sum_r w_r, with w = the number of terms this rank's mean-normalised loss
          averaged over (its local ray count: `shard_bounds` is ragged when the batch does not divide).  The weight rides in the
          last slot of the same buffer, so it is still ONE all-reduce.
        Losses normalised per SAMPLE (march_rays_train emits a different number per rank) must use the sum form."""
        if weight is not None and not average:
            raise ValueError("GradientBucket.all_reduce: weight only applies to average=True")
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            for p in self.params:   # a view was replaced (set_to_none / first backward after detach): the collective would miss it
                if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * self.flat.element_size():
                    raise RuntimeError("GradientBucket: a .grad no longer aliases the flat buffer; call attach() after zero_grad(set_to_none=True)")
            if weight is not None:
                self.flat.mul_(float(weight))
                self._buf[-1] = float(weight)
                dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.div_(self._buf[-1])   # device-side divide: no host round trip
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                if average:
                    self.flat.mul_(1.0 / dist.get_world_size(self.group))
        return self.flat


def broadcast_state(tensors, src=0, group=None):
    """replicate rank `src`'s tensors (initial weights, the 256 KB density bitfield after an occupancy update) on every rank, in place"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
