"""Spherical-harmonics direction encoder -- operator API of the reference's `shencoder` package
(/root/reference/shencoder/sphere_harmonics.py:14-86) on the gfx950 kernel (csrc/lz_encoders.hip)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from ._util import call, ptr, require_cuda, stream


class _sh_encoder(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)  # force float32 (sphere_harmonics.py:16)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        # inputs: [B, 3] float in [-1, 1] -> [B, degree^2]
        inputs = inputs.contiguous()
        B, input_dim = inputs.shape
        output_dim = degree ** 2
        require_cuda(inputs=inputs)
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, input_dim * output_dim, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else None
        call("lz_sh_encode_forward", ptr(inputs), ptr(outputs), B, input_dim, int(degree), ptr(dy_dx), stream())
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = [B, input_dim, degree]
        return outputs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        grad = grad.contiguous()
        B, input_dim, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        call("lz_sh_encode_backward", ptr(grad), ptr(inputs), B, input_dim, int(degree), ptr(dy_dx), ptr(grad_inputs), stream())
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        # inputs: [..., 3] in [-size, size] -> [..., degree^2]
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
