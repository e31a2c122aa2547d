"""Frequency (positional) encoder -- operator API of the reference's `freqencoder` package
(/root/reference/freqencoder/freq.py:15-76) on the gfx950 kernel (csrc/lz_encoders.hip).
Layout: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...], each block input_dim wide."""
import torch
import torch.nn as nn
from torch.autograd import Function

from ._util import call, ptr, require_cuda, stream


class _freq_encoder(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, output_dim):
        if not inputs.is_cuda:
            inputs = inputs.cuda()
        inputs = inputs.contiguous()
        B, input_dim = inputs.shape
        require_cuda(inputs=inputs)
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        call("lz_freq_encode_forward", ptr(inputs), B, input_dim, int(degree), int(output_dim), ptr(outputs), stream())
        ctx.save_for_backward(inputs, outputs)
        ctx.dims = [B, input_dim, degree, output_dim]
        return outputs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        grad = grad.contiguous()
        inputs, outputs = ctx.saved_tensors
        B, input_dim, degree, output_dim = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        call("lz_freq_encode_backward", ptr(grad), ptr(outputs), B, input_dim, int(degree), int(output_dim), ptr(grad_inputs), stream())
        return grad_inputs, None, None


freq_encode = _freq_encoder.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = freq_encode(inputs, self.degree, self.output_dim)
        return outputs.reshape(prefix_shape + [self.output_dim])
