"""Shared plumbing of the two per-point encoders (spherical harmonics, frequency): both are `[B, D] -> [B, C]` maps evaluated by
one HIP kernel each (csrc/lz_encoders.hip), differ only in what the backward pass needs saved, and expose the reference's
operator names through thin subclasses (shencoder.py, freqencoder.py)."""
import torch
from torch.autograd import Function

from ._util import call, ptr, require_cuda, stream

F32_FWD = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # the reference's wrappers compute in float32
F32_BWD = torch.amp.custom_bwd(device_type="cuda")


def as_rows(x, width):
    """[..., width] -> ([B, width] contiguous, leading shape)"""
    lead = tuple(x.shape[:-1])
    return x.reshape(-1, width), lead


def new_rows(like, rows, width, zero=False):
    make = torch.zeros if zero else torch.empty
    return make(rows, width, dtype=like.dtype, device=like.device)


class PointwiseOp(Function):
    """forward(ctx, x, *cfg) / backward(ctx, g) skeleton; subclasses fill `_launch_fwd` and `_launch_bwd`"""

    N_CFG = 0

    @classmethod
    def _run_forward(cls, ctx, x, out_width, want_jacobian, launch):
        x = x.contiguous()
        require_cuda(inputs=x)
        B, D = x.shape
        y = new_rows(x, B, out_width)
        jac = new_rows(x, B, D * out_width) if want_jacobian else None
        launch(x, y, jac, B, D)
        return x, y, jac


def launch(name, *args):
    call(name, *args, stream())


__all__ = ["F32_FWD", "F32_BWD", "as_rows", "new_rows", "PointwiseOp", "launch", "ptr"]
