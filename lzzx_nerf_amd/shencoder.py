"""Real spherical harmonics of a direction, degrees 1..8 -- the `shencoder` operator of the reference
(/root/reference/shencoder/sphere_harmonics.py:14-86: `sh_encode(inputs, degree, calc_grad_inputs)`, `SHEncoder(input_dim, degree)`)
evaluated by `lz_sh_encode_forward/backward` (csrc/lz_encoders.hip, generated polynomials in include/lzzx_sh_eval.h)."""
import torch.nn as nn

from ._pointwise import F32_BWD, F32_FWD, PointwiseOp, as_rows, launch, new_rows, ptr


class _sh_encoder(PointwiseOp):
    @staticmethod
    @F32_FWD
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        degree = int(degree)

        def run(x, y, jac, B, D):
            launch("lz_sh_encode_forward", ptr(x), ptr(y), B, D, degree, ptr(jac))

        x, y, jac = _sh_encoder._run_forward(ctx, inputs, degree * degree, calc_grad_inputs, run)
        ctx.save_for_backward(x, jac)
        ctx.degree = degree
        return y

    @staticmethod
    @F32_BWD
    def backward(ctx, grad):
        x, jac = ctx.saved_tensors
        if jac is None:            # directions were not asked for a gradient (sphere_harmonics.py:44-46)
            return None, None, None
        gx = new_rows(x, x.shape[0], x.shape[1], zero=True)
        grad = grad.contiguous()
        launch("lz_sh_encode_backward", ptr(grad), ptr(x), x.shape[0], x.shape[1], ctx.degree, ptr(jac), ptr(gx))
        return gx, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    """directions [..., 3] in [-size, size] -> [..., degree^2]"""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        if input_dim != 3:
            raise AssertionError("SH encoder only support input dim == 3")
        if not 1 <= degree <= 8:
            raise AssertionError("SH encoder only supports degree in [1, 8]")
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree * degree

    def extra_repr(self):
        return "input_dim=%d degree=%d" % (self.input_dim, self.degree)

    def __repr__(self):
        return "SHEncoder: " + self.extra_repr()

    def forward(self, inputs, size=1):
        rows, lead = as_rows(inputs / size, self.input_dim)
        return sh_encode(rows, self.degree, rows.requires_grad).reshape(*lead, self.output_dim)
