"""Fused triplane head for TRAINING: forward = one kernel, backward = one kernel for the whole data-gradient chain + one pass over
the per-sample records for the weight gradients + the LDS grid backward per plane.  Two arrangements of the same arithmetic:
`record=True` (default) -- the forward (`lz_triplane_head_forward_record`) writes the layer inputs and a state row per sample, the
backward (`lz_triplane_head_backward_recorded`) starts from them; `record=False` -- the forward is `lz_triplane_head_forward`, the
backward (`lz_triplane_head_backward`) recomputes the activations from (xyzs, dirs): no activation memory held between the two
(3.3 KB per sample in record mode, 1.9 KB with record_dtype="f16"), twice the matrix work in the backward.  A drop-in for the per-sample part of `NeRFNetwork.forward` in training mode (/root/reference/nerf_triplane/network.py:252-311):
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

_REC = 656   # floats per sample record of lz_triplane_head_backward (LZ_BWD_REC; slot columns: include/lzzx_nerf_hip.h LZ_BWD_*)
_STATE = 176   # floats per sample state row of lz_triplane_head_forward_record (LZ_FWD_STATE)
_REC16, _STATE16 = 704, 128   # record_dtype="f16": halves per record (LZ_BWD_REC16), dwords per state row (LZ_FWD_STATE16)
_ORDER = ["aud0", "aud1", "eye0", "eye1", "sig0", "sig1", "sig2", "col0", "col1", "unc0", "unc1"]


def _color0_output_gradient(rec, M):
    """[M, 64] f32: the (ReLU-masked) gradient of color_net.0's output from the per-sample records -- slot G_C1H of the blocked layouts
    (csrc/lz_head_bwd_common.h: f32 [slice][tile][sample 16][16 dwords]; f16: tiles interleaved in pairs, dword j of pair g =
    {tile 2 g column j | tile 2 g + 1 column j << 16}, include/lzzx_nerf_hip.h LZ_BWD_G_C1H / LZ_R16_G_C1H)"""
    n = rec.shape[0] // 16
    if rec.dtype == torch.float16:
        t = rec.view(n, _REC16 // 32, 16, 16, 2)[:, 19:21]          # pairs 19, 20 = tiles 38..41: [slice, pair, sample, column, half]
        g = t.permute(0, 2, 1, 4, 3).reshape(n * 16, 64).float()       # feature = 32 pair + 16 half + column
    else:
        t = rec.view(n, _REC // 16, 16, 16)[:, 36:40]                # tiles 36..39: [slice, tile, sample, column]
        g = t.permute(0, 2, 1, 3).reshape(n * 16, 64)
    return g[:M]


class _FusedHeadTrain(Function):
    @staticmethod
    def forward(ctx, mod, xyzs, dirs, enc_a, ind_code, eye, e_xy, e_yz, e_xz, *weights):
        dev = xyzs.device
        # opt.train_camera (renderer.py:129-132, 225-230): rays_o / rays_d carry gradients, march_rays_train hands them on to xyzs / dirs
        # and the reference reaches them through the encoders' dy_dx (grid.py:44-84, sphere_harmonics.py:27-58).  Same here, see backward
        dirs_req = bool(dirs.requires_grad)
        xyzs, dirs = xyzs.detach().float().contiguous(), dirs.detach().float().contiguous()
        M = xyzs.shape[0]
        w = [t.detach().float().contiguous() for t in weights]
        mod._pack("f32", w)
        enc_a_f = enc_a.detach().reshape(-1).float().contiguous()
        ind_f = None if ind_code is None or not mod.has_ind else ind_code.detach().reshape(-1).float().contiguous()
        eye_f = None if eye is None or not mod.has_eye else eye.detach().reshape(-1).float().contiguous()
        emb = [t.detach().float().contiguous() for t in (e_xy, e_yz, e_xz)]
        p = mod._params(emb, enc_a_f, ind_f, eye_f)
        kw = dict(dtype=torch.float32, device=dev)
        sig, rgb, aa, ae, un = torch.empty(M, **kw), torch.empty(M, 3, **kw), torch.empty(M, 1, **kw), torch.empty(M, 1, **kw), torch.empty(M, 1, **kw)
        ctx.rec = ctx.state = ctx.encx = None
        ctx.recompute = not mod.record
        Mb = (M + 15) // 16 * 16   # records and state are blocked by 16-sample slice: whole slices
        # recompute_mlp (all-f16 arrangement, round 5): the forward leaves only the enc_x halves (80 B per sample) and the backward kernel runs the
        # MLP again from them.  Gradients to the view directions need color_net.0's output gradient from a record: such a step keeps records
        use_rc = M > 0 and mod.recompute_mlp and not dirs_req
        if use_rc:
            mod._pack("f16", w)
            mod._pack("unc16", w)
            p16 = mod._params(emb, enc_a_f, ind_f, eye_f)
            p16.packed, p16.precision = mod.packed16.data_ptr(), 1
            ctx.encx = torch.empty(Mb // 16, 320, **kw)          # [slice][5][64 lanes] dwords (LZ_ENCX16_BYTES)
            call("lz_triplane_head_forward_encx_f16", C.byref(p16), ptr(mod.packed_unc16), ptr(xyzs), ptr(dirs), M, ptr(sig), ptr(rgb), ptr(aa), ptr(ae),
                 ptr(un), ptr(ctx.encx), stream())
        elif M > 0 and mod.forward_f16:
            # forward in the reference's autocast arithmetic on the f16 matrix cores (lz_head_rec16.hip); the backward below runs its
            # f32 data-gradient chain from the state this forward records
            mod._pack("f16", w)
            mod._pack("unc16", w)
            p16 = mod._params(emb, enc_a_f, ind_f, eye_f)
            p16.packed, p16.precision = mod.packed16.data_ptr(), 1
            ctx.rec, ctx.state = torch.empty(Mb, _REC16, dtype=torch.float16, device=dev), torch.empty(Mb, _STATE16, **kw)
            call("lz_triplane_head_forward_record_f16", C.byref(p16), ptr(mod.packed_unc16), ptr(xyzs), ptr(dirs), M, ptr(sig), ptr(rgb), ptr(aa),
                 ptr(ae), ptr(un), ptr(ctx.rec), ptr(ctx.state), stream())
        elif M > 0 and mod.record:
            # held until the backward consumes them (not through save_for_backward: nothing else may alias or modify them)
            try:
                if mod.record_f16:
                    ctx.rec, ctx.state = torch.empty(Mb, _REC16, dtype=torch.float16, device=dev), torch.empty(Mb, _STATE16, **kw)
                else:
                    ctx.rec, ctx.state = torch.empty(Mb, _REC, **kw), torch.empty(Mb, _STATE, **kw)
            except torch.cuda.OutOfMemoryError:
                if mod.record_f16:
                    raise
                # 3.3 KB per sample do not fit: this step runs the recomputing pair instead (same gradients, nothing held)
                ctx.rec = ctx.state = None
                ctx.recompute = True
        if use_rc:
            pass
        elif M > 0 and mod.record and not mod.forward_f16 and ctx.rec is not None:
            call("lz_triplane_head_forward_record", C.byref(p), ptr(xyzs), ptr(dirs), M, ptr(sig), ptr(rgb), ptr(aa), ptr(ae), ptr(un),
                 ptr(ctx.rec), ptr(ctx.state), int(mod.record_f16), stream())
        elif M > 0 and not mod.forward_f16:   # an empty batch (every ray missed the box) has empty outputs and zero gradients
            call("lz_triplane_head_forward", C.byref(p), ptr(xyzs), ptr(dirs), M, None, ptr(sig), ptr(rgb), ptr(aa), ptr(ae), ptr(un), stream())
        # save_for_backward (not attributes): autograd then detects an in-place update of a weight / table between forward and
        # backward (detach() shares the version counter), and the tensors are released with the graph
        opt = [t for t in (ind_f, eye_f) if t is not None]
        ctx.mod, ctx.has = mod, (ind_f is not None, eye_f is not None)
        ctx.save_for_backward(xyzs, dirs, enc_a_f, *opt, *emb, *w)
        ctx.shapes = (enc_a.shape, None if ind_code is None else ind_code.shape)
        return sig, rgb, aa, ae, un

    @staticmethod
    def backward(ctx, g_sig, g_rgb, g_aa, g_ae, g_un):
        mod = ctx.mod
        sv = list(ctx.saved_tensors)
        xyzs, dirs, enc_a_f = sv[:3]
        k = 3
        ind_f = eye_f = None
        if ctx.has[0]:
            ind_f, k = sv[k], k + 1
        if ctx.has[1]:
            eye_f, k = sv[k], k + 1
        emb, w = sv[k:k + 3], sv[k + 3:]
        M, dev = xyzs.shape[0], xyzs.device
        kw = dict(dtype=torch.float32, device=dev)
        if M == 0:
            enc_a_shape, ind_shape = ctx.shapes
            g_enc_a = torch.zeros(enc_a_shape, **kw) if ctx.needs_input_grad[3] else None
            g_ind = torch.zeros(ind_shape, **kw) if (ind_shape is not None and ctx.needs_input_grad[4] and mod.has_ind) else None
            g_x = torch.zeros(0, 3, **kw) if ctx.needs_input_grad[1] else None
            g_d = torch.zeros(0, 3, **kw) if ctx.needs_input_grad[2] else None
            return (None, g_x, g_d, g_enc_a, g_ind, None) + tuple(torch.zeros_like(t) for t in emb) + tuple(torch.zeros_like(t) for t in w)
        z = lambda g, shape: (torch.zeros(shape, **kw) if g is None else g.float().contiguous())
        g_sig, g_rgb, g_aa, g_ae, g_un = z(g_sig, (M,)), z(g_rgb, (M, 3)), z(g_aa, (M, 1)), z(g_ae, (M, 1)), z(g_un, (M, 1))
        if not ctx.recompute and ctx.rec is None and ctx.encx is None:
            # the record and state of this forward were consumed (and released) by an earlier backward
            raise RuntimeError("FusedTriplaneTrainHead(record=True): a second backward through the same forward needs record=False "
                               "(the recomputing backward keeps nothing between the two)")
        # one record per sample: every layer input / output gradient the reductions need (the X half is there already in record mode)
        rec = ctx.rec if (ctx.rec is not None or ctx.encx is not None) else torch.empty((M + 15) // 16 * 16, _REC, **kw)
        denc = torch.empty(3, 12, M, **kw)          # level-major: the grid backward reads one level at a time
        small = torch.zeros(32 + 4 + 16 + 32 + 192, **kw)   # d_enc_a | d_ind | dW of the three skinny output layers (reduced in the kernel)
        d_enc_a, d_ind, dw_e2, dw_u2, dw_c2 = small[:32], small[32:36], small[36:52], small[52:84], small[84:]
        o = _lib.HeadBwdOut()
        o.denc, o.small, o.rec = denc.data_ptr(), small.data_ptr(), (rec.data_ptr() if rec is not None else None)
        # `mod.packed` is shared by every forward of this module: re-pack from the weights THIS forward saw (a no-op when it still holds them)
        mod._pack("f32", w)
        p = mod._params(emb, enc_a_f, ind_f, eye_f)
        k_sig0, k_col0 = w[4].shape[1], w[7].shape[1]   # 68 without the eye column, 80 without an individual code
        shapes = dict(x3=(112, 36), aud1=(32, 64), sig0=(64, k_sig0), sig1=(64, 64), c1h=(65, 84))
        red = {n: torch.empty(sh, **kw) for n, sh in shapes.items()}
        if mod._gw_ws is None or mod._gw_ws.device != dev:
            mod._gw_ws = torch.empty(_lib.load().lz_triplane_head_grad_w_workspace() // 4, **kw)
        need_x, need_d = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        # d loss / d dirs starts from color_net.0's output gradient, which only the two-pass arrangements leave behind (record slot G_C1H)
        fused_dw = ctx.encx is not None or (ctx.state is not None and mod.fuse_dw and rec.dtype == torch.float16 and not need_d)
        if ctx.encx is not None:
            # the recomputing all-f16 backward: MLP forward chain again from the enc_x halves, data gradient + weight-gradient products in one kernel
            if need_d:
                raise RuntimeError("FusedTriplaneTrainHead(recompute_mlp=True): gradients to the view directions need a recorded step")
            mod._pack("f16", w)
            mod._pack("unc16", w)
            mod._pack("bwd16", w)
            o.rec = None
            call("lz_triplane_head_backward_encx_dw16", C.byref(p), ptr(mod.packed16), ptr(mod.packed_unc16), ptr(ctx.encx), ptr(dirs), M, ptr(g_sig),
                 ptr(g_rgb), ptr(g_aa), ptr(g_ae), ptr(g_un), C.byref(o), ptr(mod.packed_bwd16), k_sig0,
                 *[ptr(red[n]) for n in ("x3", "aud1", "sig0", "sig1", "c1h")], ptr(mod._gw_ws), stream())
            ctx.encx = None
        elif fused_dw:
            # the whole backward over half records in one kernel: data-gradient chain (f16 or f32 matrix path) + the weight-gradient products
            wb16 = None
            if mod.backward_f16:
                mod._pack("bwd16", w)
                wb16 = ptr(mod.packed_bwd16)
            call("lz_triplane_head_backward_recorded_dw16", C.byref(p), ptr(ctx.state), ptr(rec), M, ptr(g_sig), ptr(g_rgb), ptr(g_aa), ptr(g_ae),
                 ptr(g_un), C.byref(o), wb16, k_sig0, *[ptr(red[n]) for n in ("x3", "aud1", "sig0", "sig1", "c1h")],
                 ptr(mod._gw_ws), stream())
            ctx.state = None
        elif ctx.state is not None:
            wb16 = None
            if mod.backward_f16:   # transposed half fragments of the weights THIS forward saw
                mod._pack("bwd16", w)
                wb16 = ptr(mod.packed_bwd16)
            call("lz_triplane_head_backward_recorded", C.byref(p), ptr(ctx.state), M, ptr(g_sig), ptr(g_rgb), ptr(g_aa), ptr(g_ae), ptr(g_un),
                 C.byref(o), int(mod.record_f16), wb16, stream())
            ctx.state = None
        else:
            call("lz_triplane_head_backward", C.byref(p), ptr(xyzs), ptr(dirs), M, ptr(g_sig), ptr(g_rgb), ptr(g_aa), ptr(g_ae), ptr(g_un),
                 C.byref(o), stream())
        # weight gradients of the wide layers: ONE pass over the records (the skinny ones came out of the backward kernel)
        if not fused_dw:
            call("lz_triplane_head_grad_w_f16" if rec.dtype == torch.float16 else "lz_triplane_head_grad_w", ptr(rec), M, k_sig0,
                 *[ptr(red[n]) for n in ("x3", "aud1", "sig0", "sig1", "c1h")], ptr(mod._gw_ws), stream())
        g_dirs = None
        if need_d:
            # d SH = G_c0 . W_color0[:, SH columns]; then the SH encoder's own input gradient (shencoder.cu:358-382) from its Jacobian
            d_sh = (_color0_output_gradient(rec, M) @ w[7][:, :16]).contiguous()
            sh, jac, g_dirs = torch.empty(M, 16, **kw), torch.empty(M, 48, **kw), torch.zeros(M, 3, **kw)
            call("lz_sh_encode_forward", ptr(dirs), ptr(sh), M, 3, 4, ptr(jac), stream())
            call("lz_sh_encode_backward", ptr(d_sh), ptr(dirs), M, 3, 4, ptr(jac), ptr(g_dirs), stream())
        ctx.rec = None
        # geo = s2 . Wg^T and d geo = G_c1 . Wc[:, geo] never left the kernel: both weight gradients follow from R = sum G_c1^T s2
        x3, c1h = red["x3"], red["c1h"]
        R, w_sig2, w_col0 = c1h[:64, :64], w[6], w[7]
        d_col0 = torch.cat([c1h[:64, 64:80], R @ w_sig2[1:65].T, c1h[:64, 80:80 + (k_col0 - 80)]], 1)
        d_sig2 = torch.cat([c1h[64:65, :64], w_col0[:, 16:80].T @ R], 0)
        dws = dict(aud0=x3[:64], eye0=x3[64:80], unc0=x3[80:112], aud1=red["aud1"], sig0=red["sig0"], sig1=red["sig1"],
                   sig2=d_sig2, col0=d_col0, eye1=dw_e2.view(1, 16), unc1=dw_u2.view(1, 32), col1=dw_c2.view(3, 64))
        dws = {n: dws[n].reshape(wt.shape) for n, wt in zip(_ORDER, w)}
        # table gradients: LDS-accumulated scatter per plane, inputs mapped exactly like the forward ((x + bound) / (2 bound))
        x01 = torch.empty(3, M, 2, **kw)
        call("lz_triplane_plane_coords", ptr(xyzs), M, mod.bound, ptr(x01), stream())
        demb, dx01 = [], []
        same = all(t.shape == emb[0].shape for t in emb)
        ge_all = torch.zeros((3,) + tuple(emb[0].shape), **kw) if same else None     # one fill for the three planes' gradients
        for p_, (c, e, g) in enumerate(zip(x01, emb, denc)):
            ge = ge_all[p_] if same else torch.zeros_like(e)
            jac = gin = None
            if need_x:
                # the grid encoder's dy_dx (gridencoder.cu:179-222) is recomputed here instead of being held from the forward (96 B per
                # sample and plane); kernel_input_backward (:316-342) then runs inside lz_grid_encode_backward as on the operator path
                jac, gin, feat = torch.empty(M, 24, **kw), torch.zeros(M, 2, **kw), torch.empty(12, M, 1, **kw)
                call("lz_grid_encode_forward", ptr(c), ptr(e), ptr(mod.offsets), ptr(feat), M, 2, 1, 12, mod.S, mod.H, ptr(jac), 0, 0, 0, 0, stream())
                dx01.append(gin)
            call("lz_grid_encode_backward", ptr(g), ptr(c), ptr(e), ptr(mod.offsets), ptr(ge), M, 2, 1, 12, mod.S, mod.H, ptr(jac), ptr(gin), 0, 0,
                 0, 3 if M >= 16384 else 0, stream())
            demb.append(ge)
        if mod.keep_denc:
            mod.last_denc = denc
        g_xyzs = None
        if need_x:
            # planes xy = (x, y), yz = (y, z), xz = (x, z) (network.py:211) behind x01 = (x + bound) / (2 bound) (grid.py:143)
            g_xyzs = torch.stack([dx01[0][:, 0] + dx01[2][:, 0], dx01[0][:, 1] + dx01[1][:, 0], dx01[1][:, 1] + dx01[2][:, 1]], 1) / (2 * mod.bound)
        enc_a_shape, ind_shape = ctx.shapes
        g_enc_a = d_enc_a.view(enc_a_shape) if ctx.needs_input_grad[3] else None
        g_ind = d_ind.view(ind_shape) if (ind_shape is not None and ctx.needs_input_grad[4] and mod.has_ind) else None
        return (None, g_xyzs, g_dirs, g_enc_a, g_ind, None, demb[0], demb[1], demb[2]) + tuple(dws[n] for n in _ORDER)


class FusedTriplaneTrainHead(nn.Module):
    def __init__(self, state_dict=None, bound=1.0, exp_eye=True, ind_dim=4, record=True, record_dtype="f32", forward_dtype="f32",
                 backward_dtype="f32", fuse_dw=True, recompute_mlp=None):
        super().__init__()
        if record_dtype not in ("f32", "f16") or forward_dtype not in ("f32", "f16") or backward_dtype not in ("f32", "f16"):
            raise ValueError("record_dtype / forward_dtype / backward_dtype must be 'f32' or 'f16'")
        if forward_dtype == "f16" or backward_dtype == "f16":
            record, record_dtype = True, "f16"   # the f16 kernels record / consume half operands
        if record_dtype == "f16" and not record:
            raise ValueError("record_dtype='f16' needs record=True (the recomputing backward writes f32 records)")
        self.bound = float(bound)
        self.record = bool(record)
        # "f16": the operands of the weight-gradient products (layer inputs and output gradients) are rounded to half on their way
        # through memory, as the reference's autocast mode does for its dW GEMMs (TrainerUtil.py:103, 865-870); forward, data gradient
        # and accumulation stay f32.  Upstream gradients should come through a GradScaler like there (small ones flush to zero in half).
        self.record_f16 = record_dtype == "f16"
        # forward_dtype="f16": the forward itself runs in the reference's autocast arithmetic (half Linear inputs / weights / outputs,
        # f32 accumulate: lz_head_f16_slice.h) -- what `-O` training does there; the data gradient stays an f32 chain through the f32
        # weights, evaluated at the recorded half activations, and the weight gradients are reduced from the half records.
        self.forward_f16 = forward_dtype == "f16"
        # backward_dtype="f16": the data-gradient products on the f16 matrix cores, dY and W rounded to half, f32 sums -- the
        # reference's autocast backward; "f32" keeps the f32 chain (more accurate than autocast, 4 x the matrix instructions).
        # forward_dtype = backward_dtype = "f16" is the whole step in the arithmetic of the reference's `-O` mode.
        self.backward_f16 = backward_dtype == "f16"
        # fuse_dw (with half records): the weight gradients of the wide layers are reduced inside the backward kernel
        # (lz_triplane_head_backward_recorded_dw16) instead of in a second pass over the records, and the G half of the records is
        # never written.  Two LDS buffers of operand tiles with the f16 data gradient, one (an extra barrier per segment) with the f32 chain
        self.fuse_dw = bool(fuse_dw) and self.record_f16
        # recompute_mlp (default: on for forward_dtype = backward_dtype = "f16" with fuse_dw): the forward keeps ONLY the enc_x halves the MLP
        # started from (80 bytes per sample instead of 1 216 of record + state) and the backward kernel recomputes every layer input, mask and
        # pre-activation from them with the forward's own code (csrc/lz_head_fwd16_chain.h) -- same outputs and gradients bit for bit
        all16 = self.forward_f16 and self.backward_f16 and self.fuse_dw
        if recompute_mlp and not all16:
            raise ValueError("recompute_mlp goes with forward_dtype = backward_dtype = 'f16' and fuse_dw")
        self.recompute_mlp = all16 if recompute_mlp is None else bool(recompute_mlp)
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
        self._gw_ws = None   # partial-tile workspace of lz_triplane_head_grad_w, allocated on first backward
        # diagnostics: with keep_denc set, the last backward leaves d loss / d enc_x (level-major [3, 12, M], what the table scatter consumed)
        # in last_denc -- tests/test_gpu_train_fullsize.py checks the scatter against it (per-level sums, the checker's scatter)
        self.keep_denc, self.last_denc = False, None
        self._packed_token = {}    # weight image -> ((data_ptr, version) of the tensors it was packed from)
        if self.backward_f16:
            self.register_buffer("packed_bwd16", torch.empty(_lib.load().lz_head_packed_bwd_size_f16(), dtype=torch.uint8), persistent=False)
        if self.forward_f16:
            self.register_buffer("packed16", torch.empty(_lib.load().lz_head_packed_size_f16(), dtype=torch.uint8), persistent=False)
            self.register_buffer("packed_unc16", torch.empty(_lib.load().lz_head_packed_unc_size_f16(), dtype=torch.uint8), persistent=False)
        if state_dict is not None:
            own = self.state_dict()
            self.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items() if k in own}, strict=False)

    @property
    def offsets(self):
        return self.encoder_xy.offsets

    def invalidate_packs(self):
        """forget which tensors the weight images were packed from (call after changing a weight behind autograd's back, e.g. through
        `.data`, which bumps no version counter)"""
        self._packed_token.clear()

    def _pack(self, kind, w):
        """(re)build one of the weight images from the tensors `w` unless it already holds exactly these (same storage, same version): the
        images are shared by every forward / backward of the module, so a backward re-packs only when something else packed in between or a
        weight changed since its forward"""
        token = tuple((t.data_ptr(), t._version) for t in w) + (str(w[0].device),)
        if self._packed_token.get(kind) == token:
            return
        if kind == "f32":
            call("lz_head_pack_weights", *[ptr(t) for t in w], int(self.has_eye), int(self.has_ind), ptr(self.packed), stream())
        elif kind == "f16":
            call("lz_head_pack_weights_f16", *[ptr(t) for t in w[:9]], int(self.has_eye), int(self.has_ind), ptr(self.packed16), stream())
        elif kind == "unc16":
            call("lz_head_pack_unc_f16", ptr(w[9]), ptr(w[10]), ptr(self.packed_unc16), stream())
        elif kind == "bwd16":
            call("lz_head_pack_weights_bwd_f16", ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(w[4]), ptr(w[5]), ptr(w[6]), ptr(w[7]), int(self.has_eye),
                 int(self.has_ind), ptr(self.packed_bwd16), stream())
        else:
            raise KeyError(kind)
        self._packed_token[kind] = token

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
