"""Bias-free Linear / MLP for the training path on hand-written MFMA kernels (csrc/lz_linear.hip).

Mirror of the reference's `MLP` (/root/reference/nerf_triplane/network.py:73-94): `num_layers` bias-free nn.Linear modules in
`self.net` (same state-dict keys), ReLU between them.  The reference runs them through torch's library GEMMs; for the head's
shapes (K, N <= 84 over millions of samples) those spend 3/4 of a training step.  Here every layer is ONE kernel forward
(ReLU fused) and two backward (data gradient with the ReLU mask fused; weight gradient reduced over the samples in-kernel).
"""
import ctypes as C

import torch
import torch.nn as nn
from torch.autograd import Function

from ._util import call, ptr, require_cuda, stream


class _lz_linear(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, relu=False):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w = weight.contiguous()
        require_cuda(x=x2, weight=w)
        M, K = x2.shape
        N = w.shape[0]
        if w.shape[1] != K:
            raise RuntimeError("lz_linear: weight is [%d, %d] but the input has %d features" % (N, w.shape[1], K))
        y = torch.empty(M, N, dtype=torch.float32, device=x2.device)
        call("lz_linear_forward", ptr(x2), K, None, ptr(w), K, ptr(y), N, M, K, N, int(bool(relu)), stream())
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.relu = bool(relu)
        ctx.in_shape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = dy.reshape(-1, N).float().contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = w.t().contiguous()   # [K, N]: dX = (dY . mask) . W is the forward kernel on the transposed weight
            dx = torch.empty(M, K, dtype=torch.float32, device=x2.device)
            call("lz_linear_forward", ptr(dy2), N, ptr(y), ptr(wt), N, ptr(dx), K, M, N, K, 0, stream())
            dx = dx.view(ctx.in_shape)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            # the kernel keeps <= 24 accumulator tiles of 16 x 16 and <= 96 input columns per launch: wider layers (the torso net's
            # 116-column input) go in column blocks -- dW[:, k0:k1] only needs X[:, k0:k1] (leading dimensions stay K)
            kb = min(96, (24 // ((N + 15) // 16)) * 16)
            for k0 in range(0, K, kb):
                k1 = min(K, k0 + kb)
                call("lz_linear_grad_w", ptr(dy2), N, ptr(y), C.c_void_p(x2.data_ptr() + 4 * k0), K, C.c_void_p(dw.data_ptr() + 4 * k0), K, M,
                     k1 - k0, N, stream())
        return dx, dw, None


lz_linear = _lz_linear.apply


class MLP(nn.Module):
    """drop-in for nerf_triplane.network.MLP (network.py:73-94): same constructor, same parameters / state-dict keys"""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.net = nn.ModuleList([nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=False)
                                  for l in range(num_layers)])

    def forward(self, x):
        for l in range(self.num_layers):
            x = lz_linear(x, self.net[l].weight, l != self.num_layers - 1)
        return x
