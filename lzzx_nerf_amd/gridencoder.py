"""Multi-resolution hash / tiled grid encoder -- operator API of the reference's `gridencoder` package
(/root/reference/gridencoder/grid.py:19-154) on top of the gfx950 kernels (csrc/lz_grid.hip).

Same names, arguments, defaults, state-dict keys (`embeddings`, `offsets`) and error behaviour.  Differences
that are invisible to callers: the kernel writes [B, L*C] directly (no permute/reshape copy, grid.py:52) and
reads the gradient in that layout (no permute copy, grid.py:70); kernels run on torch's current stream.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from ._util import call, ptr, require_cuda, stream

_gridtype_to_id = {"hash": 0, "tiled": 1}


def _check_dc(D, C):
    # the reference throws std::runtime_error from the dispatch switch (gridencoder.cu:354,372)
    if C not in (1, 2, 4, 8):
        raise RuntimeError("GridEncoding: C must be 1, 2, 4, or 8.")
    if D not in (1, 2, 3, 4, 5):
        raise RuntimeError("GridEncoding: D must be 1, 2, 3, 4, or 5")


_LEVEL_SIZE_CACHE = {}


def _max_level_entries(offsets):
    """largest per-level table (entries) of an `offsets` tensor; one tiny device->host copy the first time a given
    offsets buffer is seen, cached afterwards (offsets are constants of a GridEncoder)"""
    key = (offsets.data_ptr(), offsets._version, offsets.numel())
    v = _LEVEL_SIZE_CACHE.get(key)
    if v is None:
        o = offsets.detach().cpu().numpy().astype(np.int64)
        v = int(np.max(o[1:] - o[:-1])) if o.size > 1 else 0
        if len(_LEVEL_SIZE_CACHE) > 256:
            _LEVEL_SIZE_CACHE.clear()
        _LEVEL_SIZE_CACHE[key] = v
    return v


class _grid_encode(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False):
        # inputs: [B, D] float in [0, 1]; embeddings: [sO, C]; offsets: [L + 1] int32; returns [B, L * C]
        inputs = inputs.float().contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.float32(np.log2(per_level_scale)))  # narrowed to float at the FFI, gridencoder.h:12
        H = int(base_resolution)
        _check_dc(D, C)

        # half tables only under autocast and only when C is even (grid.py:38-39)
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        embeddings = embeddings.contiguous()
        if embeddings.dtype not in (torch.float32, torch.float16):
            raise RuntimeError("embeddings must be a float32 or float16 tensor")
        if offsets.dtype != torch.int32:
            raise RuntimeError("offsets must be an int tensor")
        require_cuda(inputs=inputs, embeddings=embeddings, offsets=offsets)

        outputs = torch.empty(B, L * C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        # large batches over small tables (the triplane: <= 16384 entries per level): level-resident LDS kernel
        layout = 1
        small = D <= 3 and C <= 2 and _max_level_entries(offsets) * C * embeddings.element_size() <= 65536
        if B >= 32768 and not calc_grad_inputs and small:
            layout = 2
        call("lz_grid_encode_forward", ptr(inputs), ptr(embeddings), ptr(offsets), ptr(outputs), B, D, C, L, S, H, ptr(dy_dx),
             int(gridtype), int(bool(align_corners)), int(embeddings.dtype == torch.float16), layout, stream())

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype]
        ctx.small_levels = small
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype = ctx.dims
        grad = grad.contiguous()  # [B, L * C], consumed in place of the reference's [L, B, C] copy
        if grad.dtype != embeddings.dtype:
            grad = grad.to(embeddings.dtype)
        grad_embeddings = torch.zeros_like(embeddings)
        # small f32 tables + large batches: per-level accumulation in LDS instead of scattered global atomics
        glayout = 2 if (ctx.small_levels and B >= 16384 and embeddings.dtype == torch.float32) else 1
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        call("lz_grid_encode_backward", ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), B, D, C, L,
             S, H, ptr(dy_dx), ptr(grad_inputs), int(gridtype), int(bool(ctx.align_corners)),
             int(embeddings.dtype == torch.float16), glayout, stream())
        if dy_dx is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None


grid_encode = _grid_encode.apply


def grid_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners=False):
    """Start of every level's table (plus the total) as GridEncoder lays them out (grid.py:108-121): side = ceil(H * s^l) in
    float64 (+1 cell corner unless align_corners), entries = min(2^T, side^D) rounded up to a multiple of 8."""
    cap = 1 << log2_hashmap_size
    starts = [0]
    for level in range(num_levels):
        side = int(np.ceil(base_resolution * per_level_scale ** level)) + (0 if align_corners else 1)
        entries = min(cap, side ** input_dim)
        starts.append(starts[-1] + 8 * ((entries + 7) // 8))
    return starts


class GridEncoder(nn.Module):
    """`[..., D]` points in `[-bound, bound]` -> `[..., L * C]` features.  Parameter `embeddings` [sum of level sizes, C]
    (U(-1e-4, 1e-4)), buffer `offsets` int32 [L + 1]: the reference's state-dict contract (grid.py:91-137)."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if desired_resolution is not None:   # geometric growth that reaches desired_resolution at the last level (grid.py:95-96)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        hyper = dict(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, per_level_scale=per_level_scale,
                     log2_hashmap_size=log2_hashmap_size, base_resolution=base_resolution, gridtype=gridtype,
                     align_corners=align_corners)
        for name, value in hyper.items():
            setattr(self, name, value)
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.output_dim = num_levels * level_dim
        self.max_params = 1 << log2_hashmap_size
        starts = grid_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer("offsets", torch.tensor(starts, dtype=torch.int32))
        self.n_params = starts[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(starts[-1], level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.embeddings, -1e-4, 1e-4)

    def extra_repr(self):
        finest = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return ("input_dim=%d num_levels=%d level_dim=%d resolution=%d -> %d per_level_scale=%.4f params=%s gridtype=%s align_corners=%s"
                % (self.input_dim, self.num_levels, self.level_dim, self.base_resolution, finest, self.per_level_scale,
                   tuple(self.embeddings.shape), self.gridtype, self.align_corners))

    def __repr__(self):
        return "GridEncoder: " + self.extra_repr()

    def forward(self, inputs, bound=1):
        unit = (inputs + bound) / (2 * bound)               # [-bound, bound] -> [0, 1] (grid.py:143)
        rows = unit.view(-1, self.input_dim)
        feats = grid_encode(rows, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, rows.requires_grad,
                            self.gridtype_id, self.align_corners)
        return feats.view(*inputs.shape[:-1], self.output_dim)
