"""CPU checker for the per-sample triplane head and the render loops -- TEST INFRASTRUCTURE ONLY.

Restates, on numpy float32 arrays and the C checker kernels (oracle.py):
  * NeRFNetwork.encode_x / density / forward        /root/reference/nerf_triplane/network.py:215-223, 252-311
  * MLP (bias-free Linear + ReLU)                    network.py:73-94
  * get_rays (full-image branch)                     nerf_triplane/utils.py:226-312
  * NeRFRenderer.run_cuda_for_inference hot loop     nerf_triplane/renderer.py:476-561
  * NeRFRenderer.run_cuda training branch            renderer.py:279-304

Summation order of every Linear is explicit (`korder`, see encoders_oracle.c): the order below is the
one documented in DESIGN.md ("head arithmetic contract").  Any order is an equally valid restatement
of torch's `x @ W.T`; tests/test_golden.py pins this one against the reference's own torch modules.
"""
import numpy as np

from . import oracle as O

F32 = np.float32


# ---------------------------------------------------------------------------------------------
# summation orders
# ---------------------------------------------------------------------------------------------
def korder_natural(K, base=0):
    k = list(range(base, base + K))
    while len(k) % 4:
        k.append(-1)
    return k


def korder_chained(K, base=0):
    """Order in which a 16x16x4 f32 MFMA consumes a previous layer's accumulator tile without any
    data movement: within each block of 16 features, step r takes features 4q + r for q = 0..3."""
    Kp = (K + 15) // 16 * 16
    k = []
    for t in range(Kp // 16):
        for r in range(4):
            for q in range(4):
                f = 16 * t + 4 * q + r
                k.append(base + f if f < K else -1)
    return k


# ---------------------------------------------------------------------------------------------
# triplane hyper-parameters (network.py:129-133)
# ---------------------------------------------------------------------------------------------
class TriplaneSpec:
    def __init__(self, bound=1.0):
        self.bound = float(bound)
        self.input_dim, self.num_levels, self.level_dim = 2, 12, 1
        self.base_resolution, self.log2_hashmap_size = 64, 14
        desired = 512 * bound
        # GridEncoder.__init__, grid.py:95-96
        self.per_level_scale = np.exp2(np.log2(desired / self.base_resolution) / (self.num_levels - 1))
        self.offsets = O.grid_offsets(2, 12, self.per_level_scale, 64, 14)
        self.n_params = int(self.offsets[-1])


def encode_plane(spec, uv, emb):
    """GridEncoder.forward, grid.py:139-154: map [-bound, bound] -> [0, 1], then grid_encode."""
    x = (uv.astype(F32) + F32(spec.bound)) / F32(2 * spec.bound)
    out, _ = O.grid_encode_forward(x, emb, spec.offsets, spec.per_level_scale, spec.base_resolution)
    return out


def encode_x(spec, xyz, P):
    """network.py:208-223: xy = x[:, :2], yz = x[:, 1:], xz = x[:, [0, 2]]"""
    xyz = np.ascontiguousarray(xyz, dtype=F32)
    fxy = encode_plane(spec, xyz[:, [0, 1]], P["encoder_xy.embeddings"])
    fyz = encode_plane(spec, xyz[:, [1, 2]], P["encoder_yz.embeddings"])
    fxz = encode_plane(spec, xyz[:, [0, 2]], P["encoder_xz.embeddings"])
    return np.concatenate([fxy, fyz, fxz], axis=1)


def density(spec, P, enc_x, enc_a, eye):
    """network.py:283-311 (enc_x precomputed).  enc_a [1,32] or [32]; eye [1,1] / scalar / None."""
    enc_a = np.asarray(enc_a, dtype=F32).reshape(1, -1)
    a1 = O.linear(enc_x, P["aud_ch_att_net.net.0.weight"], korder_natural(36), relu=True)
    att = O.linear(a1, P["aud_ch_att_net.net.1.weight"], korder_chained(64))
    enc_w = enc_a * att
    parts = [enc_x, enc_w]
    order = korder_natural(36) + korder_chained(32, base=36)
    eye_att = None
    if eye is not None:
        e1 = O.linear(enc_x, P["eye_att_net.net.0.weight"], korder_natural(36), relu=True)
        e2 = O.linear_lanes(e1, P["eye_att_net.net.1.weight"])                 # VALU layer (csrc/lz_head.hip: lz_lane_dot)
        eye_att = O.unary("sigmoid", e2)
        parts.append(np.asarray(eye, dtype=F32).reshape(1, 1) * eye_att)
        order = order + [68, -1, -1, -1]
    h = np.ascontiguousarray(np.concatenate(parts, axis=1))
    s1 = O.linear(h, P["sigma_net.net.0.weight"], order, relu=True)
    s2 = O.linear(s1, P["sigma_net.net.1.weight"], korder_chained(64), relu=True)
    W3 = P["sigma_net.net.2.weight"]
    geo = O.linear(s2, np.ascontiguousarray(W3[1:]), korder_chained(64))       # 64 geo rows on the matrix cores
    sigma = O.unary("exp", np.ascontiguousarray(O.linear_lanes(s2, np.ascontiguousarray(W3[:1]))[:, 0]))   # sigma row: VALU layer
    sumsq = O.linear_lanes(att)                                                # sum of squares, lane-partial order
    amb_aud = np.sqrt(sumsq).astype(F32)
    return dict(sigma=sigma, geo_feat=geo, ambient_aud=amb_aud, ambient_eye=eye_att, enc_x=enc_x)


def head_forward(spec, P, xyz, dirs, enc_a, ind_code, eye, testing=True, unc_loss=True):
    """NeRFNetwork.forward, network.py:252-280.  Returns sigma [M], rgb [M,3], amb_aud [M,1], amb_eye [M,1]|None,
    unc [M,1] (the reference returns an over-sized [M,36,1] constant tensor in test mode, SURVEY 8a' note 16)."""
    enc_x = encode_x(spec, xyz, P)
    dres = density(spec, P, enc_x, enc_a, eye)
    enc_d, _ = O.sh_encode_forward(dirs, 4)
    parts = [enc_d, dres["geo_feat"]]
    order = korder_natural(16) + korder_chained(64, base=16)
    if ind_code is not None:
        c = np.asarray(ind_code, dtype=F32).reshape(1, -1)
        parts.append(np.repeat(c, enc_x.shape[0], axis=0))
        order = order + korder_natural(c.shape[1], base=80)
    h = np.ascontiguousarray(np.concatenate(parts, axis=1))
    c1 = O.linear(h, P["color_net.net.0.weight"], order, relu=True)
    c2 = O.linear_lanes(c1, P["color_net.net.1.weight"])                       # VALU layer
    rgb = O.unary("sigmoid", c2) * F32(1 + 2 * 0.001) - F32(0.001)
    M = enc_x.shape[0]
    if testing or not unc_loss:
        unc = np.full((M, 1), O.unary("softplus", np.zeros(1, F32))[0], dtype=F32)
    else:
        u1 = O.linear(enc_x, P["unc_net.net.0.weight"], korder_natural(36), relu=True)
        u2 = O.linear_lanes(u1, P["unc_net.net.1.weight"])                     # VALU layer
        unc = O.unary("softplus", u2)
    return dres["sigma"], rgb.astype(F32), dres["ambient_aud"], dres["ambient_eye"], unc


def head_forward_torch(spec, P, xyz, dirs, enc_a, ind_code, eye, testing=True, _cache={}):
    """NeRFNetwork.forward (network.py:252-311) with the reference's OWN arrangement of the MLPs: bias-free torch Linear stacks on
    CPU tensors (`F.linear`, in-place relu, `repeat` / `cat` materialising [M,69] and [M,84]; network.py:73-94), fp32, torch's intra-op
    thread pool; encoders = the C checker.  This is bench.py's `cpu_baseline` head ("the reference's pure-torch MLP on CPU
    tensors"); library GEMMs fix no summation order, so it agrees with head_forward to ~3e-6, not to the bit."""
    import torch
    import torch.nn.functional as F
    key = id(P)
    if key not in _cache:
        _cache.clear()
        _cache[key] = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=F32)) for k, v in P.items() if k.endswith(".weight")}
    W = _cache[key]

    def mlp(h, name, n):
        for i in range(n):
            h = F.linear(h, W[f"{name}.net.{i}.weight"])
            if i != n - 1:
                h = F.relu(h, inplace=True)
        return h

    with torch.no_grad():
        enc_x = torch.from_numpy(encode_x(spec, xyz, P))
        M = enc_x.shape[0]
        a = torch.from_numpy(np.asarray(enc_a, dtype=F32).reshape(1, -1)).repeat(M, 1)
        att = mlp(enc_x, "aud_ch_att_net", 2)
        parts = [enc_x, a * att]
        eye_att = None
        if eye is not None:
            eye_att = torch.sigmoid(mlp(enc_x, "eye_att_net", 2))
            parts.append(torch.from_numpy(np.asarray(eye, dtype=F32).reshape(1, 1)) * eye_att)
        h = mlp(torch.cat(parts, -1), "sigma_net", 3)
        sigma = torch.exp(h[..., 0])
        enc_d = torch.from_numpy(O.sh_encode_forward(dirs, 4)[0])
        cparts = [enc_d, h[..., 1:]]
        if ind_code is not None:
            cparts.append(torch.from_numpy(np.asarray(ind_code, dtype=F32).reshape(1, -1)).repeat(M, 1))
        rgb = torch.sigmoid(mlp(torch.cat(cparts, -1), "color_net", 2)) * (1 + 2 * 0.001) - 0.001
        if testing:
            unc = torch.log(1 + torch.exp(torch.zeros(M, 1)))
        else:
            unc = torch.log(1 + torch.exp(mlp(enc_x, "unc_net", 2)))
        amb_aud = att.norm(dim=-1, keepdim=True)
    n = lambda t: None if t is None else np.ascontiguousarray(t.numpy(), dtype=F32)
    return n(sigma), n(rgb), n(amb_aud), n(eye_att), n(unc)


# ---------------------------------------------------------------------------------------------
# the same forward under torch autocast (opt.fp16): what the reference computes when it renders in half precision
# ---------------------------------------------------------------------------------------------
F16 = np.float16


def _h(x):
    return np.asarray(x, dtype=F32).astype(F16)


def _lin16(x16, W):
    """autocast nn.Linear (bias-free): input and weight cast to half, f32 accumulation, half output"""
    return (x16.astype(F32) @ _h(W).astype(F32).T).astype(F16)


def head_forward_fp16(spec, P, xyz, dirs, enc_a, ind_code, eye, testing=True, enc_a_half=True, trace=None):
    """NeRFNetwork.forward (network.py:252-311) with CUDA autocast enabled (`torch.cuda.amp.autocast`, TrainerUtil.py:455,535,649,858):
    the grid encoders stay f32 (C = 1, grid.py:38), SH is f32 (custom_fwd cast_inputs), every Linear casts its input and weight to
    half and returns half (f32 accumulation), relu / sigmoid / half products run in half (f32 internally, rounded to half), `cat`
    promotes to the widest input, `exp`, `norm` and `log` are on autocast's fp32 list (half -> f32 in, f32 out).
    `enc_a_half`: the conditioning feature is half, as `encode_audio` returns it inside the autocast region (renderer.py:241); with
    False it is f32 and `enc_a * att` is an f32 product (type promotion) that only sigma_net's first Linear rounds to half.
    PINNED (tests/test_golden_autocast.py) to the reference's own Python run under torch autocast, layer by layer
    (tests/golden/reference_autocast.npz, make_golden_autocast.py).  The summation order inside a half GEMM is the library's (here
    numpy's f32 matmul), so agreement with any other implementation is to half rounding, not to the bit.
    `trace` (dict): filled with the half output of every Linear under the reference's module names."""
    relu = lambda a: np.maximum(a, F16(0))
    tr = trace if trace is not None else {}

    def mlp16(x16, name, n):
        for i in range(n):
            x16 = _lin16(x16, P[f"{name}.net.{i}.weight"])
            tr[f"{name}.net.{i}"] = x16
            if i != n - 1:
                x16 = relu(x16)
        return x16

    enc_x = encode_x(spec, xyz, P)
    M = enc_x.shape[0]
    x16 = _h(enc_x)
    att = mlp16(x16, "aud_ch_att_net", 2)
    if enc_a_half:
        enc_w = (_h(enc_a).reshape(1, -1).astype(F32) * att.astype(F32)).astype(F16)          # half * half -> half
    else:
        enc_w = _h(np.asarray(enc_a, dtype=F32).reshape(1, -1) * att.astype(F32))             # f32 * half -> f32, rounded by the Linear's cast
    parts = [x16, enc_w]
    eye_att = None
    if eye is not None:
        e2 = mlp16(x16, "eye_att_net", 2)
        eye_att = _h(O.unary("sigmoid", np.ascontiguousarray(e2.astype(F32))))
        parts.append(_h(np.asarray(eye, dtype=F32).reshape(1, 1) * eye_att.astype(F32)))     # f32 [1,1] * half -> f32 -> cast by the Linear
    s3 = mlp16(np.concatenate(parts, axis=1), "sigma_net", 3)
    sigma = O.unary("exp", np.ascontiguousarray(s3[:, 0].astype(F32)))                        # fp32 list
    enc_d, _ = O.sh_encode_forward(dirs, 4)
    parts = [_h(enc_d), s3[:, 1:]]
    if ind_code is not None:
        parts.append(np.repeat(_h(ind_code).reshape(1, -1), M, axis=0))
    c2 = mlp16(np.concatenate(parts, axis=1), "color_net", 2)
    sg = _h(O.unary("sigmoid", np.ascontiguousarray(c2.astype(F32))))
    rgb = _h(_h(sg.astype(F32) * F32(1 + 2 * 0.001)).astype(F32) - F32(0.001)).astype(F32)
    amb_aud = np.sqrt((att.astype(F32) ** 2).sum(1, keepdims=True)).astype(F32)              # norm: fp32 list
    if testing:
        unc = np.full((M, 1), O.unary("softplus", np.zeros(1, F32))[0], dtype=F32)           # zeros_like(enc_x): f32
    else:
        u2 = mlp16(x16, "unc_net", 2)
        unc = O.unary("softplus", np.ascontiguousarray(u2.astype(F32)))                       # exp, log: fp32 list
    return sigma, rgb, amb_aud, None if eye_att is None else eye_att.astype(F32), unc


# ---------------------------------------------------------------------------------------------
# rays
# ---------------------------------------------------------------------------------------------
def get_rays(pose, intrinsics, H, W):
    """utils.py:226-312, N = -1 branch, one pose [4,4] -> rays_o, rays_d [H*W, 3] (C checker lzo_get_rays)."""
    return O.get_rays(pose, intrinsics, H, W)
