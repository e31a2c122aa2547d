"""Fused triplane head for TRAINING: forward = `lz_k_triplane_head<true>` (one kernel), backward = `lz_triplane_head_backward`
(one kernel for the whole data-gradient chain, activations recomputed) + `lz_linear_grad_w` per layer + the LDS grid backward per
plane.  A drop-in for the per-sample part of `NeRFNetwork.forward` in training mode (/root/reference/nerf_triplane/network.py:252-311):
same parameters, same state-dict keys, same five outputs.

    net = FusedTriplaneTrainHead(state_dict, bound=1.0)
    sigma, rgb, amb_aud, amb_eye, unc = net(xyzs, dirs, enc_a, ind_code, eye)      # autograd-ready
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from ._util import call, ptr, stream
from .gridencoder import GridEncoder
from .linear import MLP

_X = dict(X_encx=36, X_a1=64, X_e1=16, X_sig0=72, X_s1=64, X_s2=64, X_col0=84, X_c1=64, X_u1=32)
_G = dict(G_a1=64, G_att=32, G_e1=16, G_e2=1, G_s1=64, G_s2=64, G_s3=65, G_c1=64, G_c=3, G_u1=32, G_u=1)
# layer -> (X buffer, its leading dimension, K, G buffer, N)
_LAYERS = [("aud0", "X_encx", 36, 36, "G_a1", 64), ("aud1", "X_a1", 64, 64, "G_att", 32), ("eye0", "X_encx", 36, 36, "G_e1", 16),
           ("eye1", "X_e1", 16, 16, "G_e2", 1), ("sig0", "X_sig0", 72, 69, "G_s1", 64), ("sig1", "X_s1", 64, 64, "G_s2", 64),
           ("sig2", "X_s2", 64, 64, "G_s3", 65), ("col0", "X_col0", 84, 84, "G_c1", 64), ("col1", "X_c1", 64, 64, "G_c", 3),
           ("unc0", "X_encx", 36, 36, "G_u1", 32), ("unc1", "X_u1", 32, 32, "G_u", 1)]
_ORDER = ["aud0", "aud1", "eye0", "eye1", "sig0", "sig1", "sig2", "col0", "col1", "unc0", "unc1"]


class _FusedHeadTrain(Function):
    @staticmethod
    def forward(ctx, mod, xyzs, dirs, enc_a, ind_code, eye, e_xy, e_yz, e_xz, *weights):
        dev = xyzs.device
        xyzs, dirs = xyzs.detach().float().contiguous(), dirs.detach().float().contiguous()
        M = xyzs.shape[0]
        w = [t.detach().float().contiguous() for t in weights]
        call("lz_head_pack_weights", *[ptr(t) for t in w], int(mod.has_eye), int(mod.has_ind), ptr(mod.packed), stream())
        enc_a_f = enc_a.detach().reshape(-1).float().contiguous()
        ind_f = None if ind_code is None or not mod.has_ind else ind_code.detach().reshape(-1).float().contiguous()
        eye_f = None if eye is None or not mod.has_eye else eye.detach().reshape(-1).float().contiguous()
        emb = [t.detach().float().contiguous() for t in (e_xy, e_yz, e_xz)]
        p = mod._params(emb, enc_a_f, ind_f, eye_f)
        kw = dict(dtype=torch.float32, device=dev)
        sig, rgb, aa, ae, un = torch.empty(M, **kw), torch.empty(M, 3, **kw), torch.empty(M, 1, **kw), torch.empty(M, 1, **kw), torch.empty(M, 1, **kw)
        call("lz_triplane_head_forward", C.byref(p), ptr(xyzs), ptr(dirs), M, None, ptr(sig), ptr(rgb), ptr(aa), ptr(ae), ptr(un), stream())
        ctx.mod, ctx.saved = mod, (xyzs, dirs, enc_a_f, ind_f, eye_f, emb, w)
        ctx.shapes = (enc_a.shape, None if ind_code is None else ind_code.shape)
        return sig, rgb, aa, ae, un

    @staticmethod
    def backward(ctx, g_sig, g_rgb, g_aa, g_ae, g_un):
        mod = ctx.mod
        xyzs, dirs, enc_a_f, ind_f, eye_f, emb, w = ctx.saved
        M, dev = xyzs.shape[0], xyzs.device
        kw = dict(dtype=torch.float32, device=dev)
        z = lambda g, shape: (torch.zeros(shape, **kw) if g is None else g.float().contiguous())
        g_sig, g_rgb, g_aa, g_ae, g_un = z(g_sig, (M,)), z(g_rgb, (M, 3)), z(g_aa, (M, 1)), z(g_ae, (M, 1)), z(g_un, (M, 1))
        widths = {**_X, **_G}
        names = list(widths)
        al = lambda n: (n + 3) // 4 * 4                                         # every buffer starts 16-byte aligned (dwordx4 stores)
        work = torch.empty(sum(al(M * wd) for wd in widths.values()), **kw)     # one allocation for every dump buffer
        bufs, off = {}, 0
        for n in names:
            bufs[n] = work[off: off + M * widths[n]].view(M, widths[n])
            off += al(M * widths[n])
        denc = [torch.empty(12, M, **kw) for _ in range(3)]   # level-major: the grid backward reads one level at a time
        d_enc_a, d_ind = torch.zeros(32, **kw), torch.zeros(4, **kw)
        o = _lib.HeadBwdOut()
        for i in range(3):
            o.denc[i] = denc[i].data_ptr()
        o.d_enc_a, o.d_ind = d_enc_a.data_ptr(), d_ind.data_ptr()
        for n in names:
            setattr(o, n, bufs[n].data_ptr())
        p = mod._params(emb, enc_a_f, ind_f, eye_f)
        call("lz_triplane_head_backward", C.byref(p), ptr(xyzs), ptr(dirs), M, ptr(g_sig), ptr(g_rgb), ptr(g_aa), ptr(g_ae), ptr(g_un),
             C.byref(o), stream())
        # weight gradients: one in-kernel reduction over the M samples per layer
        dws = {}
        for (name, xb, ldx, K, gb, N), wt in zip(_LAYERS, w):
            if not mod.has_eye and name.startswith("eye"):
                dws[name] = torch.zeros_like(wt)
                continue
            dw = torch.zeros_like(wt)
            call("lz_linear_grad_w", ptr(bufs[gb]), widths[gb], None, ptr(bufs[xb]), ldx, ptr(dw), wt.shape[1], M, wt.shape[1], N, stream())
            dws[name] = dw
        # table gradients: LDS-accumulated scatter per plane, inputs mapped exactly like the forward ((x + bound) / (2 bound))
        demb = []
        for cols, e, g in zip(((0, 1), (1, 2), (0, 2)), emb, denc):
            x01 = ((xyzs[:, list(cols)] + mod.bound) / (2 * mod.bound)).contiguous()
            ge = torch.zeros_like(e)
            call("lz_grid_encode_backward", ptr(g), ptr(x01), ptr(e), ptr(mod.offsets), ptr(ge), M, 2, 1, 12, mod.S, mod.H, None, None, 0, 0,
                 0, 3 if M >= 16384 else 0, stream())
            demb.append(ge)
        enc_a_shape, ind_shape = ctx.shapes
        g_enc_a = d_enc_a.view(enc_a_shape) if ctx.needs_input_grad[3] else None
        g_ind = d_ind.view(ind_shape) if (ind_shape is not None and ctx.needs_input_grad[4] and mod.has_ind) else None
        return (None, None, None, g_enc_a, g_ind, None, demb[0], demb[1], demb[2]) + tuple(dws[n] for n in _ORDER)


class FusedTriplaneTrainHead(nn.Module):
    def __init__(self, state_dict=None, bound=1.0, exp_eye=True, ind_dim=4):
        super().__init__()
        self.bound = float(bound)
        mk = lambda: GridEncoder(input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=14,
                                 desired_resolution=512 * bound)
        self.encoder_xy, self.encoder_yz, self.encoder_xz = mk(), mk(), mk()                       # network.py:131-133
        self.has_eye, self.has_ind = bool(exp_eye), ind_dim > 0
        self.aud_ch_att_net = MLP(36, 32, 64, 2)
        self.eye_att_net = MLP(36, 1, 16, 2)
        self.sigma_net = MLP(36 + 32 + (1 if exp_eye else 0), 65, 64, 3)
        self.color_net = MLP(16 + 64 + ind_dim, 3, 64, 2)
        self.unc_net = MLP(36, 1, 32, 2)
        self.H = 64
        self.S = float(np.float32(np.log2(self.encoder_xy.per_level_scale)))
        self.register_buffer("packed", torch.empty(_lib.load().lz_head_packed_size(), dtype=torch.float32), persistent=False)
        if state_dict is not None:
            own = self.state_dict()
            self.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items() if k in own}, strict=False)

    @property
    def offsets(self):
        return self.encoder_xy.offsets

    def _weights(self):
        n = lambda m, i: m.net[i].weight
        return [n(self.aud_ch_att_net, 0), n(self.aud_ch_att_net, 1), n(self.eye_att_net, 0), n(self.eye_att_net, 1), n(self.sigma_net, 0),
                n(self.sigma_net, 1), n(self.sigma_net, 2), n(self.color_net, 0), n(self.color_net, 1), n(self.unc_net, 0), n(self.unc_net, 1)]

    def _params(self, emb, enc_a, ind_code, eye):
        p = _lib.HeadParams()
        p.emb_xy, p.emb_yz, p.emb_xz = [t.data_ptr() for t in emb]
        p.offsets, p.packed, p.enc_a = self.offsets.data_ptr(), self.packed.data_ptr(), enc_a.data_ptr()
        p.ind_code = None if ind_code is None else ind_code.data_ptr()
        p.eye = None if eye is None else eye.data_ptr()
        p.bound, p.S, p.H, p.testing, p.precision = self.bound, self.S, self.H, 0, 0
        return p

    def forward(self, xyzs, dirs, enc_a, ind_code=None, eye=None):
        return _FusedHeadTrain.apply(self, xyzs, dirs, enc_a, ind_code, eye, self.encoder_xy.embeddings, self.encoder_yz.embeddings,
                                     self.encoder_xz.embeddings, *self._weights())
