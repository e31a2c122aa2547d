"""BASELINE cfg2 on fused kernels: a generic hash-grid NeRF on the operators of `encoding.get_encoder` (encoding.py:6-37) --
hashgrid defaults (input_dim 3, num_levels 16, level_dim 2, log2_hashmap_size 19), SH(4) directions, a sigma MLP 32-64-16 and a colour
MLP 31-64-3 (bias-free Linear + ReLU like network.py:73-94; sigma = exp(row 0), rgb = sigmoid) -- as

    FusedHashgridNeRF   the per-sample network: level-major gather (tiled output, never untiled) + ONE MFMA kernel for both MLPs, SH and
                        the activations (csrc/lz_ngp.hip) -- what `GridEncoder -> MLP -> cat -> MLP` does in ~25 launches per iteration
    HashgridRenderer    the reference's inference loop (renderer.py:495-548) around it: loop state on the device, 4 launches per
                        iteration (march, gather, head, composite) enqueued by one C call per chunk, no host round trip

Same operators, same arithmetic per sample as the operator-API network (`synthetic.GenericHashgridNeRF`), up to the summation order of
the Linear layers (here: MFMA k order, restated by the checker in oracle/ngp.py; there: the lz_linear kernels')."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._util import call, ptr, stream
from .renderer import MAX_RAYS_PER_PASS, _STATE_INTS, _N_SAMPLES_OFF   # noqa: F401

NGP_FRAGS = 96          # LZ_NGP_FRAGS
TILE_ROWS = 256         # LZ_GRID_TILE_ROWS
GEO = 15                # geometry features handed from the sigma net to the colour net
HIDDEN = 64


def _fragment_tables():
    """(layer, row, col, keep) per packed float: fragment (ks, ft), lane l = W[16 ft + (l & 15)][k(ks, l >> 4)] (csrc/lz_ngp.hip)"""
    lane = np.arange(64)
    m, q = lane & 15, lane >> 4
    layer, row, col, keep = [], [], [], []

    def add(which, r, c, k):
        layer.append(np.full(64, which)); row.append(r); col.append(c); keep.append(k)

    one = np.ones(64, bool)
    for ks in range(8):                     # sigma_net.0 [64, 32]: lane q reads levels q, q + 4, q + 8, q + 12, both channels
        for ft in range(4):
            add(0, 16 * ft + m, 2 * (q + 4 * (ks >> 1)) + (ks & 1), one)
    for ks in range(16):                    # sigma_net.1 [16, 64]: the previous accumulator tile in place
        add(1, m, 16 * (ks >> 2) + 4 * q + (ks & 3), one)
    for ks in range(8):                     # colour_net.0 [64, 31]: SH component 4 ks + q, then sigma_net output 4 q + r (row 0 = sigma: weight 0)
        for ft in range(4):
            if ks < 4:
                add(2, 16 * ft + m, 4 * ks + q, one)
            else:
                slot = 4 * q + (ks - 4)
                add(2, 16 * ft + m, np.maximum(16 + slot - 1, 0), slot >= 1)
    for ks in range(16):                    # colour_net.1 [3, 64] in one 16-row tile
        add(3, np.minimum(m, 2), 16 * (ks >> 2) + 4 * q + (ks & 3), m < 3)
    return [np.concatenate(a) for a in (layer, row, col, keep)]


_TABLES = None


def pack_weights(sigma_w0, sigma_w1, color_w0, color_w1):
    """the four weight matrices (torch Linear layout [out, in]) -> LZ_NGP_FRAGS * 64 floats in MFMA fragment order, on their device"""
    global _TABLES
    shapes = [(HIDDEN, 32), (1 + GEO, HIDDEN), (HIDDEN, 16 + GEO), (3, HIDDEN)]
    ws = [w.detach().float() for w in (sigma_w0, sigma_w1, color_w0, color_w1)]
    for w, sh in zip(ws, shapes):
        if tuple(w.shape) != sh:
            raise RuntimeError("FusedHashgridNeRF supports sigma MLP 32-64-16 and colour MLP 31-64-3 (got a weight of shape %s where %s is expected); "
                               "other architectures run on the operator API (renderer.NetworkRenderer)" % (tuple(w.shape), sh))
    if _TABLES is None:
        _TABLES = _fragment_tables()
    layer, row, col, keep = _TABLES
    dev = ws[0].device
    out = torch.zeros(NGP_FRAGS * 64, dtype=torch.float32, device=dev)
    for i, w in enumerate(ws):
        sel = np.nonzero((layer == i) & keep)[0]
        out[torch.from_numpy(sel).to(dev)] = w[torch.from_numpy(row[sel]).to(dev), torch.from_numpy(col[sel]).to(dev)]
    return out.contiguous()


class FusedHashgridNeRF:
    """encoder: a gridencoder.GridEncoder built by get_encoder('hashgrid') (input_dim 3, level_dim 2, gridtype hash, no align_corners);
    sigma_net / color_net: linear.MLP or anything with `.net[i].weight` (two bias-free layers each).  half_tables: gather from an f16
    copy of the table and hand f16 features to the head (what grid.py:28,38-39 does under autocast with an even level_dim)."""

    def __init__(self, encoder, sigma_net, color_net, half_tables=False):
        if encoder.input_dim != 3 or encoder.level_dim != 2 or encoder.gridtype_id != 0 or encoder.align_corners or encoder.num_levels != 16:
            raise RuntimeError("FusedHashgridNeRF: the encoder must be get_encoder('hashgrid') with input_dim 3, num_levels 16, level_dim 2")
        self.encoder = encoder
        self.half_tables = bool(half_tables)
        self.packed = pack_weights(sigma_net.net[0].weight, sigma_net.net[1].weight, color_net.net[0].weight, color_net.net[1].weight)
        emb = encoder.embeddings.detach()
        self.table = emb.half().contiguous() if self.half_tables else emb.float().contiguous()
        self.offsets = encoder.offsets.contiguous()
        self.S = float(np.log2(encoder.per_level_scale))

    def forward(self, xyzs, dirs, bound=1.0, out=None):
        """xyzs [M, 3] in [-bound, bound], dirs [M, 3] -> (sigma [M], rgb [M, 3]); ready in stream order"""
        xyzs = xyzs.reshape(-1, 3).float().contiguous()
        dirs = dirs.reshape(-1, 3).float().contiguous()
        M, dev = xyzs.shape[0], xyzs.device
        feats = torch.empty(M, 32, dtype=self.table.dtype, device=dev)
        sigma, rgb = out if out is not None else (torch.empty(M, device=dev), torch.empty(M, 3, device=dev))
        self.encode_tiled(xyzs, feats, bound)
        call("lz_ngp_head_forward", ptr(self.packed), ptr(feats), 2 if self.half_tables else 1, ptr(dirs), M, None, ptr(sigma), ptr(rgb), stream())
        return sigma, rgb

    def encode_tiled(self, xyzs, feats, bound, count_ptr=None):
        e = self.encoder
        call("lz_grid_encode_forward_tiled", ptr(xyzs), ptr(self.table), ptr(self.offsets), ptr(feats), xyzs.shape[0], count_ptr, float(bound), 3, 2,
             e.num_levels, self.S, e.base_resolution, 0, 0, int(self.half_tables), stream())


class HashgridRenderer:
    """Inference renderer of one FusedHashgridNeRF + occupancy bitfield: the reference's loop (renderer.py:495-548: march n_step samples per
    alive ray, network, composite_rays, drop dead rays, n_step = max(min(budget_factor * N // n_alive, n_step_cap), 1), stop at max_steps)
    with the state on the device; (budget_factor, n_step_cap) = (1, 8) is the reference's schedule -- same pixels under any schedule for
    rays that end before max_steps, and exactly the reference's cap semantics under (1, 8), the default (like TriplaneRenderer /
    NetworkRenderer; fatter schedules such as (8, 8) are faster and differ on rays that reach max_steps)."""

    def __init__(self, net, density_bitfield, bound=1.0, cascade=None, grid_size=128, aabb=None, min_near=0.05, budget_factor=1, n_step_cap=8):
        import math
        self.net = net
        self.bound = float(bound)
        self.cascade = cascade if cascade is not None else 1 + math.ceil(math.log2(bound))   # renderer.py:93
        self.grid_size = grid_size
        self.bitfield = density_bitfield.contiguous()
        dev = self.bitfield.device
        if aabb is None:
            aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32, device=dev)
        self.aabb = aabb.to(dev, torch.float32).contiguous()
        self.min_near = float(min_near)
        self.budget_factor, self.n_step_cap = int(budget_factor), int(n_step_cap)
        self.chunk, self.lookahead = 4, 2
        self._buf = None

    def _buffers(self, N, dev):
        rows = max(N * self.budget_factor, N)
        b = self._buf
        if b is None or b["N"] != N or b["rows"] != rows or b["dev"] != dev:
            f = dict(dtype=torch.float32, device=dev)
            i = dict(dtype=torch.int32, device=dev)
            b = dict(N=N, rows=rows, dev=dev, nears=torch.empty(N, **f), fars=torch.empty(N, **f), rays_alive=[torch.empty(N, **i), torch.empty(N, **i)],
                     rays_t=torch.empty(N, **f), weights_sum=torch.empty(N, **f), depth=torch.empty(N, **f), image=torch.empty(N, 3, **f),
                     out=torch.empty(N, 3, **f), xyzs=torch.zeros(rows, 3, **f), dirs=torch.zeros(rows, 3, **f), deltas=torch.zeros(rows, 2, **f),
                     feats=torch.zeros(rows, 32, dtype=self.net.table.dtype, device=dev), sigmas=torch.empty(rows, **f), rgbs=torch.empty(rows, 3, **f),
                     state=torch.zeros(_STATE_INTS, **i), workspace=torch.empty(4096, **i), ray_counts=torch.zeros(N, **i),
                     ring=[torch.zeros(8, dtype=torch.int32).pin_memory() for _ in range(4)])
            self._buf = b
        return b

    @torch.no_grad()
    def render(self, rays_o, rays_d, dt_gamma=1.0 / 256, max_steps=128, T_thresh=1e-4, bg_color=1.0, count_samples=False):
        """-> dict(image [N,3] blended + clamped, image_raw, weights_sum, depth, state, ray_counts if requested); ready in stream order"""
        rays_o = rays_o.reshape(-1, 3).float().contiguous()
        rays_d = rays_d.reshape(-1, 3).float().contiguous()
        N, dev = rays_o.shape[0], rays_o.device
        if N > MAX_RAYS_PER_PASS:
            raise RuntimeError("HashgridRenderer renders at most %d rays per call" % MAX_RAYS_PER_PASS)
        if N == 0:
            from .renderer import _empty_result
            return _empty_result(dev, count_samples, ambient=False)
        b = self._buffers(N, dev)
        call("lz_near_far_from_aabb", ptr(rays_o), ptr(rays_d), ptr(self.aabb), N, self.min_near, ptr(b["nears"]), ptr(b["fars"]), stream())
        if count_samples:
            b["ray_counts"].zero_()
        call("lz_loop_begin", N, int(max_steps), N * self.budget_factor, self.n_step_cap, ptr(b["nears"]), ptr(b["rays_alive"][0]), ptr(b["rays_t"]),
             ptr(b["weights_sum"]), ptr(b["depth"]), ptr(b["image"]), None, None, None, ptr(b["state"]), ptr(b["workspace"]), stream())
        net, e = self.net, self.net.encoder
        f = _lib.FrameNgp()
        p = lambda t: t.data_ptr()
        f.packed, f.embeddings, f.offsets, f.feats = p(net.packed), p(net.table), p(net.offsets), p(b["feats"])
        f.enc_L, f.enc_H, f.enc_S, f.emb_f16 = e.num_levels, e.base_resolution, net.S, int(net.half_tables)
        f.state, f.workspace = p(b["state"]), p(b["workspace"])
        f.rays_alive[0], f.rays_alive[1] = p(b["rays_alive"][0]), p(b["rays_alive"][1])
        f.rays_t, f.rays_o, f.rays_d, f.nears, f.fars, f.grid = p(b["rays_t"]), p(rays_o), p(rays_d), p(b["nears"]), p(b["fars"]), p(self.bitfield)
        f.xyzs, f.dirs, f.deltas, f.sigmas, f.rgbs = p(b["xyzs"]), p(b["dirs"]), p(b["deltas"]), p(b["sigmas"]), p(b["rgbs"])
        f.weights_sum, f.depth, f.image = p(b["weights_sum"]), p(b["depth"]), p(b["image"])
        f.ray_counts = p(b["ray_counts"]) if count_samples else None
        f.N, f.max_steps, f.C, f.H = N, int(max_steps), int(self.cascade), int(self.grid_size)
        f.bound, f.dt_gamma, f.T_thresh = self.bound, float(dt_gamma), float(T_thresh)
        f.sample_budget, f.n_step_cap = N * self.budget_factor, self.n_step_cap
        limit = int(max_steps) + 1            # n_step >= 1; the state of iteration k is committed by iteration k + 1's launches
        cur, it, pending = 0, 0, []
        while it < limit:
            n = min(self.chunk, limit - it)
            call("lz_ngp_loop_run", C.byref(f), cur, n, stream())
            cur = (cur + n) & 1
            it += n
            k = len(pending)
            slot = b["ring"][k % len(b["ring"])]
            slot.copy_(b["state"][:8], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending.append((ev, slot))
            if k >= self.lookahead:       # look at the chunk `lookahead` chunks back: the queue never drains, few no-op chunks are enqueued
                ev_old, slot_old = pending[k - self.lookahead]
                ev_old.synchronize()
                if int(slot_old[3]) == 1:
                    break
        bg = bg_color.to(dev, torch.float32).expand(N, 3).contiguous() if torch.is_tensor(bg_color) else None
        call("lz_final_blend", ptr(b["image"]), ptr(b["weights_sum"]), ptr(bg), 1.0 if bg is not None else float(bg_color), N, ptr(b["out"]), stream())
        self._keep = (bg, rays_o, rays_d)
        res = dict(image=b["out"], image_raw=b["image"], weights_sum=b["weights_sum"], depth=b["depth"], state=b["state"])
        if count_samples:
            res["ray_counts"] = b["ray_counts"]
        return res
