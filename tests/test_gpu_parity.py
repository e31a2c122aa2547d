"""GPU parity: every gfx950 kernel (through the C ABI, via the operator wrappers) against the CPU checker on the
same seeded inputs.  Integer / index outputs must be bit-exact; float outputs are asserted bit-exact too wherever
both sides evaluate the same IEEE operation sequence (that is the design: explicit FMAs, deterministic exp/sin,
f32 MFMA == fma chain), and to a stated tolerance where a reduction order is free (atomics) or libm is involved.
"""
import os
import numpy as np
import pytest
import torch

from conftest import ellipsoid_bitfield, synthetic_camera
from oracle import oracle as O
from oracle.head import TriplaneSpec, get_rays, head_forward
from oracle.render import render_inference, render_train_forward

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _native_loaded():
    from lzzx_nerf_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert _lib.load().lz_device_ok() == 1, "liblzzx_nerf_hip.so needs a gfx950 device"
    yield


# ------------------------------------------------------------------------------------------------
# grid encoder
# ------------------------------------------------------------------------------------------------
GRID_CASES = [
    # D, L, C, H, log2T, desired, gridtype
    (2, 12, 1, 64, 14, 512, "hash"),     # one triplane plane (network.py:131)
    (3, 16, 2, 16, 19, 2048, "hash"),    # get_encoder('hashgrid') defaults = BASELINE cfg2
    (2, 16, 2, 16, 16, 2048, "tiled"),   # torso encoder (network.py:166)
    (1, 4, 4, 8, 10, 64, "hash"),
    (3, 6, 8, 8, 12, 128, "tiled"),
    (4, 3, 2, 4, 12, 16, "hash"),
    (5, 2, 1, 4, 12, 8, "hash"),
    (3, 16, 2, 16, 19, None, "hash"),    # GridEncoder() class defaults: per_level_scale 2 -> res up to 2^19, uint32 strides wrap
    (2, 16, 1, 16, 19, None, "hash"),
    (2, 16, 2, 16, 12, None, "tiled"),
]


@pytest.mark.parametrize("D,L,C,H,T,res,gt", GRID_CASES)
def test_grid_forward_bit_exact(D, L, C, H, T, res, gt):
    from lzzx_nerf_amd.gridencoder import GridEncoder, grid_encode
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=T, desired_resolution=res,
                      gridtype=gt).cuda()
    rng = np.random.default_rng(D * 100 + L)
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float32)
    enc.embeddings.data.copy_(dev(emb))
    B = 5003
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2, 0] = 1.0000001  # out of range
    x[3, -1] = -1e-7
    off = host(enc.offsets)
    assert np.array_equal(off, O.grid_offsets(D, L, enc.per_level_scale, H, T))
    gid = 0 if gt == "hash" else 1
    out_o, dd_o = O.grid_encode_forward(x, emb, off, enc.per_level_scale, H, True, gid)
    xt = dev(x)
    out = grid_encode(xt, enc.embeddings, enc.offsets, enc.per_level_scale, H, True, gid, False)
    assert np.array_equal(host(out), out_o)
    out_sm = grid_encode(xt, enc.embeddings, enc.offsets, enc.per_level_scale, H, False, gid, False)  # hot (sample-major) kernel
    assert np.array_equal(host(out_sm), out_o)
    # indices, bit for bit
    from lzzx_nerf_amd._util import call, ptr, stream
    idx = torch.empty(L, B, 1 << D, dtype=torch.int32, device="cuda")
    call("lz_grid_corner_indices", ptr(xt), ptr(enc.offsets), ptr(idx), B, D, C, L, float(np.float32(np.log2(enc.per_level_scale))),
         H, gid, 0, stream())
    assert np.array_equal(host(idx), O.grid_corner_indices(x, off, C, enc.per_level_scale, H, gid))
    # level-major layout of the reference FFI + dy_dx
    out_lm = torch.empty(L, B, C, device="cuda")
    dd = torch.empty(B, L * D * C, device="cuda")
    call("lz_grid_encode_forward", ptr(xt), ptr(enc.embeddings.data), ptr(enc.offsets), ptr(out_lm), B, D, C, L,
         float(np.float32(np.log2(enc.per_level_scale))), H, ptr(dd), gid, 0, 0, 0, stream())
    assert np.array_equal(host(out_lm.permute(1, 0, 2).reshape(B, L * C)), out_o)
    assert np.array_equal(host(dd), dd_o)


@pytest.mark.parametrize("D,L,C,H,T,res,gt,half", [
    (2, 12, 1, 64, 14, 512, "hash", False),     # every level fits LDS (<= 64 KB)
    (3, 16, 2, 16, 19, 2048, "hash", False),    # fine levels do not fit: in-kernel fallback to global gathers
    (3, 16, 2, 16, 19, 2048, "hash", True),
    (2, 16, 2, 16, 16, 2048, "tiled", False),   # tiled wrap: generic modulo path
    (3, 16, 2, 16, 19, None, "hash", False),    # wrapped uint32 strides
    (1, 6, 1, 8, 10, 256, "hash", False),
])
def test_grid_forward_level_resident_kernel_bit_exact(D, L, C, H, T, res, gt, half):
    """B >= 32768 selects lz_k_grid_forward_lds (level table resident in LDS, XCD-grouped workgroups)"""
    from lzzx_nerf_amd.gridencoder import GridEncoder, grid_encode
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=T, desired_resolution=res,
                      gridtype=gt).cuda()
    rng = np.random.default_rng(77)
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float16 if half else np.float32)
    B = 40000 + 123  # ragged last chunk
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[:3] = [[0.0] * D, [1.0] * D, [1.0000001] + [0.5] * (D - 1)]
    gid = 0 if gt == "hash" else 1
    out_o, _ = O.grid_encode_forward(x, emb, host(enc.offsets), enc.per_level_scale, H, False, gid)
    from lzzx_nerf_amd._util import call, ptr, stream
    xt, et = dev(x), dev(emb)
    out = torch.empty(B, L * C, device="cuda", dtype=et.dtype)
    call("lz_grid_encode_forward", ptr(xt), ptr(et), ptr(enc.offsets), ptr(out), B, D, C, L, float(np.float32(np.log2(enc.per_level_scale))),
         H, None, gid, 0, int(half), 2, stream())   # out_layout 2: force the level-resident kernel, fitting or not
    auto = grid_encode(xt, et, enc.offsets, enc.per_level_scale, H, False, gid, False)  # wrapper's own choice of kernel
    assert torch.equal(auto, out)
    if half:
        assert np.array_equal(host(out).view(np.uint16), out_o.view(np.uint16))
    else:
        assert np.array_equal(host(out), out_o)


@pytest.mark.parametrize("D,L,C,H,T,res,gt,half", [
    (3, 16, 2, 16, 19, 2048, "hash", False),    # BASELINE cfg2: 4 MB levels, 256-sample tiles
    (3, 16, 2, 16, 19, 2048, "hash", True),
    (2, 12, 1, 64, 14, 512, "hash", False),     # C = 1: one dword per (sample, level)
    (2, 16, 2, 16, 16, 2048, "tiled", False),   # tiled wrap: generic modulo path
    (3, 16, 2, 16, 19, None, "hash", False),    # wrapped uint32 strides
    (3, 6, 8, 8, 12, 128, "tiled", False),      # 8 dwords per group
    (3, 16, 8, 16, 14, 512, "hash", False),     # L*W = 128 -> 128-sample tiles
    (2, 32, 8, 4, 10, 256, "hash", False),      # L*W = 256 -> 64-sample tiles
    (3, 16, 4, 16, 15, 512, "hash", True),
])
def test_grid_forward_tiled_level_major_bit_exact(D, L, C, H, T, res, gt, half):
    """B >= 65536 with out_layout 1 selects lz_k_grid_forward_lm + lz_k_grid_untile (level-major tiles, untiled in place)"""
    from lzzx_nerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=T, desired_resolution=res,
                      gridtype=gt).cuda()
    rng = np.random.default_rng(78)
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float16 if half else np.float32)
    B = 65536 + 256 + 77  # ragged last tile for every tile size
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[:3] = [[0.0] * D, [1.0] * D, [1.0000001] + [0.5] * (D - 1)]
    x[-1] = [-1e-7] + [0.25] * (D - 1)
    gid = 0 if gt == "hash" else 1
    out_o, _ = O.grid_encode_forward(x, emb, host(enc.offsets), enc.per_level_scale, H, False, gid)
    from lzzx_nerf_amd._util import call, ptr, stream
    xt, et = dev(x), dev(emb)
    out = torch.full((B, L * C), 7.0, device="cuda", dtype=et.dtype)
    call("lz_grid_encode_forward", ptr(xt), ptr(et), ptr(enc.offsets), ptr(out), B, D, C, L, float(np.float32(np.log2(enc.per_level_scale))),
         H, None, gid, 0, int(half), 1, stream())
    if half:
        assert np.array_equal(host(out).view(np.uint16), out_o.view(np.uint16))
    else:
        assert np.array_equal(host(out), out_o)


@pytest.mark.parametrize("D,L,C,H,T,res,gt", [
    (2, 12, 1, 64, 14, 512, "hash"),     # one triplane plane: every level <= 64 KB
    (3, 8, 2, 16, 13, 256, "hash"),      # 8192 x 2 floats = 64 KB exactly
    (2, 16, 2, 16, 12, 2048, "tiled"),
    (3, 8, 2, 16, 16, 256, "hash"),      # fine levels do not fit: in-kernel fallback to global atomics
])
def test_grid_backward_level_resident(D, L, C, H, T, res, gt):
    """grad_layout 2 / 3: lz_k_grid_backward_lds_fx (per-level 64-bit fixed-point accumulation in LDS, contiguous flush) vs the checker and vs the
    plain scatter kernel; the summation order is free, so tolerance, not bits"""
    from lzzx_nerf_amd.gridencoder import GridEncoder
    from lzzx_nerf_amd._util import call, ptr, stream
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=T, desired_resolution=res,
                      gridtype=gt).cuda()
    rng = np.random.default_rng(79)
    B = 50000 + 17
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[:3] = [[0.0] * D, [1.0] * D, [1.0000001] + [0.5] * (D - 1)]
    g = rng.normal(size=(B, L * C)).astype(np.float32)
    gid = 0 if gt == "hash" else 1
    off = host(enc.offsets)
    ge, _ = O.grid_encode_backward(g, x, tuple(enc.embeddings.shape), off, enc.per_level_scale, H, None, gid)
    xt, gt_ = dev(x), dev(g)
    S = float(np.float32(np.log2(enc.per_level_scale)))
    res_ = {}
    for layout in (1, 2, 3):
        gemb = torch.zeros_like(enc.embeddings.data)
        gin = gt_ if layout != 3 else gt_.view(B, L, C).permute(1, 0, 2).contiguous()
        call("lz_grid_encode_backward", ptr(gin), ptr(xt), ptr(enc.embeddings.data), ptr(enc.offsets), ptr(gemb), B, D, C, L, S, H,
             None, None, gid, 0, 0, layout, stream())
        res_[layout] = host(gemb)
        scale = np.abs(ge).max()
        assert np.max(np.abs(res_[layout] - ge)) < 1e-5 * scale * 10, layout
    # autograd wrapper picks the level-resident kernel for large B over small tables
    enc.embeddings.grad = None
    out = enc(dev(x * 2 - 1), bound=1)
    out.backward(gt_)
    if T <= 14:
        assert np.max(np.abs(host(enc.embeddings.grad) - ge)) < 1e-4 * np.abs(ge).max()


@pytest.mark.parametrize("D,L,C,H,T,res,gt", [
    (2, 16, 2, 16, 16, 2048, "tiled"),   # the torso encoder (network.py:166): the table that goes half under `-O` training
    (3, 8, 2, 16, 15, 512, "hash"),
    (3, 4, 4, 8, 12, 64, "hash"),        # two half2 atomics per corner
    (2, 6, 8, 8, 10, 128, "tiled"),
    (2, 12, 1, 64, 14, 512, "hash"),     # C odd: the at::Half CAS branch (never reached through grid.py:38, kept correct)
])
def test_grid_backward_half_tables(D, L, C, H, T, res, gt):
    """emb_f16 = 1: `__half2` atomics of half(w * g) into a half table (gridencoder.cu:296-311) against the checker's restatement
    (oracle/grid_oracle.c:lzo_grid_encode_backward_f16).  Atomic order is free: entries with <= 2 terms must match bit for bit (half
    addition commutes), the others stay inside the rounding budget of their own terms, (terms - 1) half-ulps of sum |term|.
    grad_inputs (Half += Half * Half over (level, channel), gridencoder.cu:316-342) is sequential: bit for bit."""
    from lzzx_nerf_amd.gridencoder import GridEncoder
    from lzzx_nerf_amd._util import call, ptr, stream
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=T, desired_resolution=res,
                      gridtype=gt).cuda()
    rng = np.random.default_rng(81 + D + C)
    B = 3000 + 11
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[:3] = [[0.0] * D, [1.0] * D, [1.0000001] + [0.5] * (D - 1)]
    x[3:40] = x[40]                                       # 38 samples in ONE cell: entries with many terms
    g = (rng.normal(size=(B, L * C)) * rng.choice([1e-3, 1.0, 30.0], size=(B, 1))).astype(np.float16)
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float16)
    gid = 0 if gt == "hash" else 1
    off = host(enc.offsets)
    _, dd = O.grid_encode_forward(x, emb, off, enc.per_level_scale, H, True, gid)
    r = O.grid_encode_backward_f16(g, x, tuple(emb.shape), off, enc.per_level_scale, H, dd, gid)
    S = float(np.float32(np.log2(enc.per_level_scale)))
    xt, et, ddt = dev(x), dev(emb), dev(dd)
    terms, exact, absum = r["terms"], r["exact"], r["absum"]
    assert terms.max() >= 38 and ((terms == 1) | (terms == 2)).sum() > 50
    for layout in (0, 1):
        gin = dev(g) if layout == 1 else dev(np.ascontiguousarray(g.reshape(B, L, C).transpose(1, 0, 2)))
        gemb = torch.zeros_like(et)
        ginp = torch.zeros(B, D, dtype=torch.float16, device="cuda")
        call("lz_grid_encode_backward", ptr(gin), ptr(xt), ptr(et), ptr(enc.offsets), ptr(gemb), B, D, C, L, S, H, ptr(ddt), ptr(ginp),
             gid, 0, 1, layout, stream())
        ge = host(gemb)
        few = terms <= 2
        assert np.array_equal(ge[few].view(np.uint16), r["grad_embeddings"][few].view(np.uint16)), layout
        err = np.abs(ge.astype(np.float64) - exact)
        budget = np.maximum(terms - 1, 0) * 2.0 ** -11 * np.maximum(absum, 2.0 ** -14) + 2.0 ** -11 * np.abs(exact) + 2.0 ** -25
        assert np.all(err <= budget), (layout, float((err / budget).max()))
        assert np.array_equal(host(ginp).view(np.uint16), r["grad_inputs"].view(np.uint16)), layout
    # the operator under autocast: _grid_encode.forward casts an even-C table to half (grid.py:38-39) and backward returns its gradient
    if C % 2 == 0:
        from lzzx_nerf_amd.gridencoder import grid_encode
        enc.embeddings.data.copy_(dev(emb).float())
        with torch.autocast("cuda", dtype=torch.float16):
            out = grid_encode(xt, enc.embeddings, enc.offsets, enc.per_level_scale, H, False, gid, False)
        assert out.dtype == torch.float16
        out_o, _ = O.grid_encode_forward(x, emb, off, enc.per_level_scale, H, False, gid)
        assert np.array_equal(host(out).view(np.uint16), out_o.view(np.uint16))
        out.backward(dev(g))
        gg = host(enc.embeddings.grad)
        assert gg.dtype == np.float32                       # autograd hands the parameter its own dtype back
        few = terms <= 2
        assert np.array_equal(gg[few].astype(np.float16).view(np.uint16), r["grad_embeddings"][few].view(np.uint16))


GRID_AC_CASES = [
    (2, 12, 1, 64, 14, 512, "hash"),     # a triplane plane with align_corners
    (3, 16, 2, 16, 19, 2048, "hash"),    # cfg2's table
    (2, 16, 2, 16, 16, 2048, "tiled"),
    (3, 5, 4, 8, 12, 64, "tiled"),
    (1, 4, 8, 8, 10, 64, "hash"),
]


@pytest.mark.parametrize("D,L,C,H,T,res,gt", GRID_AC_CASES)
def test_grid_align_corners_bit_exact(D, L, C, H, T, res, gt):
    """align_corners=True (encoding.py:9,25,29 -> gridencoder.cu:62,135: no half-cell offset, side = resolution): table layout, corner
    indices, values and dy_dx through every forward kernel, and the backward, against the checker (whose branch is pinned to
    grid_sample(align_corners=True) by tests/test_oracle_known_answers.py)"""
    from lzzx_nerf_amd.gridencoder import GridEncoder, grid_encode
    from lzzx_nerf_amd._util import call, ptr, stream
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=T, desired_resolution=res,
                      gridtype=gt, align_corners=True).cuda()
    off = host(enc.offsets)
    assert np.array_equal(off, O.grid_offsets(D, L, enc.per_level_scale, H, T, align_corners=True))
    assert not np.array_equal(off, O.grid_offsets(D, L, enc.per_level_scale, H, T))
    rng = np.random.default_rng(300 + D * 10 + C)
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float32)
    enc.embeddings.data.copy_(dev(emb))
    gid = 0 if gt == "hash" else 1
    S = float(np.float32(np.log2(enc.per_level_scale)))
    for B in (5003, 70000 + 19):           # small: sample-major / dy_dx kernels; large: level-resident and tiled level-major kernels
        x = rng.uniform(0, 1, (B, D)).astype(np.float32)
        x[:4] = [[0.0] * D, [1.0] * D, [1.0000001] + [0.5] * (D - 1), [0.5] * (D - 1) + [-1e-7]]
        out_o, dd_o = O.grid_encode_forward(x, emb, off, enc.per_level_scale, H, True, gid, True)
        xt = dev(x)
        out = grid_encode(xt, enc.embeddings, enc.offsets, enc.per_level_scale, H, True, gid, True)
        assert np.array_equal(host(out), out_o)
        out2 = grid_encode(xt, enc.embeddings, enc.offsets, enc.per_level_scale, H, False, gid, True)     # the wrapper's kernel choice
        assert np.array_equal(host(out2), out_o)
        for layout in (0, 1, 2):
            o = torch.full((L, B, C) if layout == 0 else (B, L * C), 7.0, device="cuda")
            dd = torch.empty(B, L * D * C, device="cuda") if layout == 0 else None
            call("lz_grid_encode_forward", ptr(xt), ptr(enc.embeddings.data), ptr(enc.offsets), ptr(o), B, D, C, L, S, H, ptr(dd), gid, 1, 0,
                 layout, stream())
            got = host(o.permute(1, 0, 2).reshape(B, L * C)) if layout == 0 else host(o)
            assert np.array_equal(got, out_o), (B, layout)
            if dd is not None:
                assert np.array_equal(host(dd), dd_o)
        idx = torch.empty(L, B, 1 << D, dtype=torch.int32, device="cuda")
        call("lz_grid_corner_indices", ptr(xt), ptr(enc.offsets), ptr(idx), B, D, C, L, S, H, gid, 1, stream())
        assert np.array_equal(host(idx), O.grid_corner_indices(x, off, C, enc.per_level_scale, H, gid, True))
        # it is a different function of x than the default convention
        out_d, _ = O.grid_encode_forward(x[:64], emb, off, enc.per_level_scale, H, False, gid, False)
        assert not np.array_equal(out_d, out_o[:64])
        # backward: every layout (plain atomics, level-resident LDS accumulation where it applies) + grad_inputs
        g = rng.normal(size=(B, L * C)).astype(np.float32)
        ge, gi = O.grid_encode_backward(g, x, tuple(emb.shape), off, enc.per_level_scale, H, dd_o, gid, True)
        gt_, ddt = dev(g), dev(dd_o)
        for layout in (0, 1, 2, 3):
            gin = gt_ if layout in (1, 2) else gt_.view(B, L, C).permute(1, 0, 2).contiguous()
            gemb = torch.zeros_like(enc.embeddings.data)
            ginp = torch.zeros(B, D, device="cuda")
            call("lz_grid_encode_backward", ptr(gin), ptr(xt), ptr(enc.embeddings.data), ptr(enc.offsets), ptr(gemb), B, D, C, L, S, H,
                 ptr(ddt), ptr(ginp), gid, 1, 0, layout, stream())
            assert np.max(np.abs(host(gemb) - ge)) < 1e-4 * np.abs(ge).max(), (B, layout)
            assert np.array_equal(host(ginp), gi), (B, layout)
    # the module end to end (GridEncoder.forward maps [-bound, bound] -> [0, 1] and passes self.align_corners, grid.py:143-150)
    xm = rng.uniform(-1, 1, (777, D)).astype(np.float32)
    u = (xm + np.float32(1)) / np.float32(2)
    ref, _ = O.grid_encode_forward(u, emb, off, enc.per_level_scale, H, False, gid, True)
    assert np.array_equal(host(enc(dev(xm), bound=1)), ref)


def test_grid_backward_level_resident_propagates_nan():
    """a NaN (or inf) output gradient reaches grad_embeddings like the reference's float atomicAdd (gridencoder.cu:226-313): GradScaler's
    non-finite check looks at exactly these tensors.  The fixed-point LDS path cannot represent it, so the (level, chunk) is rerouted."""
    from lzzx_nerf_amd.gridencoder import GridEncoder
    from lzzx_nerf_amd._util import call, ptr, stream
    enc = GridEncoder(input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=14, desired_resolution=512).cuda()
    rng = np.random.default_rng(80)
    B, L = 40000, 12
    x = rng.uniform(0, 1, (B, 2)).astype(np.float32)
    g = rng.normal(size=(B, L)).astype(np.float32)
    g[1234, 5] = np.nan
    g[777, 0] = np.inf
    S = float(np.float32(np.log2(enc.per_level_scale)))
    off = host(enc.offsets)
    idx = O.grid_corner_indices(x, off, 1, enc.per_level_scale, 64, 0)   # [L, B, 4] entry indices incl. the level offset
    for layout in (2, 3):
        gin = dev(g) if layout == 2 else dev(np.ascontiguousarray(g.T))
        gemb = torch.zeros_like(enc.embeddings.data)
        call("lz_grid_encode_backward", ptr(gin), ptr(dev(x)), ptr(enc.embeddings.data), ptr(enc.offsets), ptr(gemb), B, 2, 1, L, S, 64,
             None, None, 0, 0, 0, layout, stream())
        ge = host(gemb)[:, 0]
        assert np.isnan(ge[idx[5, 1234]]).all(), layout
        assert not np.isfinite(ge[idx[0, 777]]).any(), layout
        other = np.ones(ge.shape[0], bool)
        other[off[5]:off[6]] = False
        other[off[0]:off[1]] = False
        assert np.isfinite(ge[other]).all(), layout


def test_grid_forward_half_tables_bit_exact():
    from lzzx_nerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=8, level_dim=2, base_resolution=16, log2_hashmap_size=15, desired_resolution=512).cuda()
    rng = np.random.default_rng(5)
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float16)
    enc.embeddings.data.copy_(dev(emb.astype(np.float32)))
    x = rng.uniform(-1, 1, (4001, 3)).astype(np.float32)
    with torch.autocast("cuda", dtype=torch.float16):
        out = enc(dev(x), bound=1)
    assert out.dtype == torch.float16  # C even + autocast -> half tables (grid.py:38-39)
    x01 = (x + np.float32(1)) / np.float32(2)
    out_o, _ = O.grid_encode_forward(x01, emb, host(enc.offsets), enc.per_level_scale, 16)
    assert np.array_equal(host(out).view(np.uint16), out_o.view(np.uint16))
    # C odd -> stays float32 under autocast (grid.py:38)
    enc1 = GridEncoder(input_dim=2, num_levels=4, level_dim=1, base_resolution=16, log2_hashmap_size=12, desired_resolution=128).cuda()
    with torch.autocast("cuda", dtype=torch.float16):
        assert enc1(dev(x[:, :2])).dtype == torch.float32


def test_grid_backward_autograd():
    from lzzx_nerf_amd.gridencoder import GridEncoder
    for (D, L, C, H, T, res) in [(2, 12, 1, 64, 14, 512), (3, 8, 2, 16, 14, 256)]:
        enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=T, desired_resolution=res).cuda()
        rng = np.random.default_rng(6)
        emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float32)
        enc.embeddings.data.copy_(dev(emb))
        x = rng.uniform(-1, 1, (3000, D)).astype(np.float32)
        xt = dev(x).requires_grad_(True)
        out = enc(xt, bound=1)
        g = rng.normal(size=tuple(out.shape)).astype(np.float32)
        out.backward(dev(g))
        x01 = (x + np.float32(1)) / np.float32(2)
        _, dd = O.grid_encode_forward(x01, emb, host(enc.offsets), enc.per_level_scale, H, True)
        ge, gi = O.grid_encode_backward(g, x01, emb.shape, host(enc.offsets), enc.per_level_scale, H, dd)
        # scatter-add order is free (atomics): tolerance, not bits
        assert np.allclose(host(enc.embeddings.grad), ge, atol=2e-4, rtol=1e-4)
        assert np.allclose(host(xt.grad), gi / 2, atol=1e-2, rtol=1e-3)  # chain rule through (x + bound) / (2 bound)
        # linearity: <g, f(emb)> == <grad_emb, emb>
        assert float((dev(g) * out).sum()) == pytest.approx(float((enc.embeddings.grad * enc.embeddings.data).sum()), rel=1e-3)


def test_grid_errors_and_edge_cases():
    from lzzx_nerf_amd.gridencoder import grid_encode
    off = torch.tensor([0, 16], dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="C must be"):
        grid_encode(torch.rand(4, 2, device="cuda"), torch.zeros(16, 3, device="cuda"), off, 2.0, 2)
    with pytest.raises(RuntimeError, match="D must be"):
        grid_encode(torch.rand(4, 6, device="cuda"), torch.zeros(16, 2, device="cuda"), off, 2.0, 2)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        grid_encode(torch.rand(4, 2), torch.zeros(16, 2), off.cpu(), 2.0, 2)
    out = grid_encode(torch.rand(0, 2, device="cuda"), torch.zeros(16, 2, device="cuda"), off, 2.0, 2)
    assert tuple(out.shape) == (0, 2)


# ------------------------------------------------------------------------------------------------
# SH / freq
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("degree", list(range(1, 9)))
def test_sh_bit_exact(degree):
    from lzzx_nerf_amd.shencoder import SHEncoder
    rng = np.random.default_rng(degree)
    d = rng.normal(size=(3001, 3)).astype(np.float32)
    d[:1500] /= np.linalg.norm(d[:1500], axis=1, keepdims=True)
    enc = SHEncoder(degree=degree)
    dt = dev(d).requires_grad_(True)
    out = enc(dt)
    out_o, dd_o = O.sh_encode_forward(d, degree, True)
    assert np.array_equal(host(out), out_o)
    g = rng.normal(size=out_o.shape).astype(np.float32)
    out.backward(dev(g))
    assert np.array_equal(host(dt.grad), O.sh_encode_backward(g, dd_o, degree))
    with torch.autocast("cuda", dtype=torch.float16):
        assert enc(dev(d).half()).dtype == torch.float32  # cast_inputs=float32 (sphere_harmonics.py:16)


def test_freq_bit_exact():
    from lzzx_nerf_amd.freqencoder import FreqEncoder
    rng = np.random.default_rng(0)
    for D, deg in ((2, 8), (6, 3), (3, 6)):
        x = rng.uniform(-1, 1, (2003, D)).astype(np.float32)
        enc = FreqEncoder(input_dim=D, degree=deg)
        xt = dev(x).requires_grad_(True)
        out = enc(xt)
        out_o = O.freq_encode_forward(x, deg)
        assert np.array_equal(host(out), out_o)
        g = rng.normal(size=out_o.shape).astype(np.float32)
        out.backward(dev(g))
        assert np.array_equal(host(xt.grad), O.freq_encode_backward(g, out_o, D, deg))


# ------------------------------------------------------------------------------------------------
# raymarching utilities
# ------------------------------------------------------------------------------------------------
def _camera_rays(H, W, rot=0.0):
    pose, intr = synthetic_camera(H, W)
    if rot:
        R = np.array([[np.cos(rot), 0, np.sin(rot)], [0, 1, 0], [-np.sin(rot), 0, np.cos(rot)]], dtype=np.float32)
        pose[:3, :3] = R
        pose[:3, 3] = R @ np.array([0, 0, -3.35], dtype=np.float32)
    return pose, intr, get_rays(pose, intr, H, W)


def test_get_rays_and_near_far_bit_exact():
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd.renderer import get_rays as gpu_get_rays
    for rot in (0.0, 0.4):
        pose, intr, _ = _camera_rays(96, 80, rot)
        intr = [intr[0] / 3, intr[1] / 3, intr[2], intr[3]]  # wide field of view: some rays miss the box
        ro, rd = get_rays(pose, intr, 96, 80)
        ro_g, rd_g = gpu_get_rays(dev(pose), intr, 96, 80)
        assert np.array_equal(host(ro_g), ro) and np.array_equal(host(rd_g), rd)
        aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
        n_o, f_o = O.near_far_from_aabb(ro, rd, aabb, 0.05)
        n_g, f_g = R.near_far_from_aabb(ro_g, rd_g, dev(aabb), 0.05)
        assert np.array_equal(host(n_g), n_o) and np.array_equal(host(f_g), f_o)
        assert (n_o > 1e30).any() and (n_o < 1e30).any()


def test_morton_packbits_dilation_sph():
    from lzzx_nerf_amd import raymarching as R
    rng = np.random.default_rng(1)
    coords = rng.integers(0, 128, (100000, 3)).astype(np.int32)
    idx = R.morton3D(dev(coords))
    assert idx.dtype == torch.int32 and np.array_equal(host(idx), O.morton3D(coords))
    assert np.array_equal(host(R.morton3D_invert(idx)), coords)
    grid = rng.uniform(0, 20, (2, 64 ** 3)).astype(np.float32)
    assert np.array_equal(host(R.packbits(dev(grid), 10.0)), O.packbits(grid, 10.0))
    bf = torch.zeros(2 * 64 ** 3 // 8, dtype=torch.uint8, device="cuda")
    assert R.packbits(dev(grid), 3.0, bf).data_ptr() == bf.data_ptr()  # caller-supplied buffer (raymarching.py:147-150)
    assert np.array_equal(host(R.morton3D_dilation(dev(grid))), O.morton3D_dilation(grid))
    _, _, (ro, rd) = _camera_rays(32, 32)
    c = host(R.sph_from_ray(dev(ro), dev(rd), 4.0))
    assert np.allclose(c, O.sph_from_ray(ro, rd, 4.0), atol=2e-6)  # atan2f / sqrtf from libm vs ocml


# ------------------------------------------------------------------------------------------------
# marching
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scene", ["ones", "ellipsoid"])
@pytest.mark.parametrize("bound,cascade", [(1.0, 1), (2.0, 2)])
def test_march_train_bit_exact(scene, bound, cascade):
    from lzzx_nerf_amd import raymarching as R
    _, _, (ro, rd) = _camera_rays(48, 48, 0.2)
    aabb = np.array([-bound, -bound / 2, -bound, bound, bound / 2, bound], np.float32)
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.05)
    if scene == "ones":
        bits = np.full(cascade * 128 ** 3 // 8, 255, np.uint8)
    else:
        b1, _ = ellipsoid_bitfield()
        bits = np.concatenate([b1] * cascade)
    N = ro.shape[0]
    noises = np.random.default_rng(3).uniform(0, 1, N).astype(np.float32)
    ctr_o = np.zeros(2, np.int32)
    xo, do, lo, ro_ = O.march_rays_train(ro, rd, bound, bits, cascade, 128, nears, fars, ctr_o, -1, noises, 128, True, 1 / 256, 48)
    # perturb=False path through the public wrapper (noise = 0), then the noisy path through the C ABI
    ctr = torch.zeros(2, dtype=torch.int32, device="cuda")
    xg, dg, lg, rg = R.march_rays_train(dev(ro), dev(rd), bound, dev(bits), cascade, 128, dev(nears), dev(fars), ctr, -1, False, 128,
                                        True, 1 / 256, 48)
    x0, d0, l0, r0 = O.march_rays_train(ro, rd, bound, bits, cascade, 128, nears, fars, np.zeros(2, np.int32), -1, None, 128, True,
                                        1 / 256, 48)
    assert np.array_equal(host(rg), r0) and np.array_equal(host(xg), x0) and np.array_equal(host(dg), d0) and np.array_equal(host(lg), l0)
    assert host(ctr).tolist() == [int(r0[:, 2].sum()), N]
    from lzzx_nerf_amd._util import call, ptr, stream
    M = xo.shape[0]
    xt, dt_, lt = torch.zeros(M, 3, device="cuda"), torch.zeros(M, 3, device="cuda"), torch.zeros(M, 2, device="cuda")
    rt = torch.empty(N, 3, dtype=torch.int32, device="cuda")
    ctr.zero_()
    ws = torch.empty(N + 2, dtype=torch.int32, device="cuda")
    keep = [dev(a) for a in (ro, rd, bits, nears, fars, noises)]  # raw pointers: the tensors must outlive the launch
    call("lz_march_rays_train", ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), bound, 1 / 256, 48, N, cascade, 128, M, ptr(keep[3]),
         ptr(keep[4]), ptr(xt), ptr(dt_), ptr(lt), ptr(rt), ptr(ctr), ptr(keep[5]), ptr(ws), stream())
    assert np.array_equal(host(rt), ro_) and np.array_equal(host(xt), xo) and np.array_equal(host(lt), lo)
    assert host(ctr).tolist() == ctr_o.tolist()


def test_march_train_mean_count_overflow_and_backward():
    from lzzx_nerf_amd import raymarching as R
    _, _, (ro, rd) = _camera_rays(16, 16)
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.05)
    bits = np.full(128 ** 3 // 8, 255, np.uint8)
    ctr = torch.zeros(2, dtype=torch.int32, device="cuda")
    rot, rdt = dev(ro).requires_grad_(True), dev(rd).requires_grad_(True)
    xg, dg, lg, rg = R.march_rays_train(rot, rdt, 1.0, dev(bits), 1, 128, dev(nears), dev(fars), ctr, 1000, False, 128, False, 1 / 256, 32)
    xo, do, lo, ro_ = O.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, np.zeros(2, np.int32), 1000, None, 128, False, 1 / 256, 32)
    assert xg.shape[0] == 1024 and np.array_equal(host(rg), ro_) and np.array_equal(host(xg), xo)
    gx = np.random.default_rng(0).normal(size=(1024, 3)).astype(np.float32)
    gd = np.random.default_rng(1).normal(size=(1024, 3)).astype(np.float32)
    torch.autograd.backward([xg, dg], [dev(gx), dev(gd)])
    go, gdd = O.march_rays_train_backward(gx, gd, ro_, lo)
    assert np.array_equal(host(rot.grad), go) and np.array_equal(host(rdt.grad), gdd)


def test_march_inference_bit_exact_and_padding_rule():
    from lzzx_nerf_amd import raymarching as R
    _, _, (ro, rd) = _camera_rays(40, 40, -0.3)
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.05)
    bits, _ = ellipsoid_bitfield()
    N = ro.shape[0]
    rng = np.random.default_rng(2)
    alive = rng.permutation(N)[: N // 2].astype(np.int32)
    rays_t = (nears + rng.uniform(0, 0.5, N).astype(np.float32)).astype(np.float32)
    for n_step in (1, 3, 8):
        xo, do, lo = O.march_rays(len(alive), n_step, alive, rays_t, ro, rd, 1.0, bits, 1, 128, nears, fars, 128, None, 1 / 256, 64)
        xg, dg, lg = R.march_rays(len(alive), n_step, dev(alive), dev(rays_t), dev(ro), dev(rd), 1.0, dev(bits), 1, 128, dev(nears),
                                  dev(fars), 128, False, 1 / 256, 64)
        M = len(alive) * n_step
        assert xg.shape[0] == M + 128 - M % 128  # always adds (raymarching.py:381-382)
        assert np.array_equal(host(xg), xo) and np.array_equal(host(dg), do) and np.array_equal(host(lg), lo)
    xg, _, _ = R.march_rays(128, 1, dev(alive[:128]), dev(rays_t), dev(ro), dev(rd), 1.0, dev(bits), 1, 128, dev(nears), dev(fars), 128)
    assert xg.shape[0] == 256


# ------------------------------------------------------------------------------------------------
# compositing
# ------------------------------------------------------------------------------------------------
def _train_inputs(seed=0, N=300, max_c=40):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, max_c, N)
    counts[:5] = 0
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    rays = np.stack([rng.permutation(N), offs, counts], 1).astype(np.int32)
    M = int(counts.sum()) + 64
    rays[-1, 2] = 200  # overflows M -> treated as empty (raymarching.cu:1904)
    sig = rng.uniform(0, 80, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    dl = np.stack([rng.uniform(0.005, 0.03, M), np.cumsum(rng.uniform(0.01, 0.03, M)) + 2], 1).astype(np.float32)
    a0, a1, unc = [rng.uniform(0, 1, M).astype(np.float32) for _ in range(3)]
    return rays, sig, rgb, dl, a0, a1, unc, M, N


@pytest.mark.parametrize("variant", ["ambient", "sigma", "uncertainty", "triplane"])
def test_composite_train_bit_exact(variant):
    from lzzx_nerf_amd import raymarching as R
    rays, sig, rgb, dl, a0, a1, unc, M, N = _train_inputs()
    t = lambda a: dev(a).requires_grad_(True)
    sg, rg, a0g, a1g, ug = t(sig), t(rgb), t(a0), t(a1), t(unc)
    rng = np.random.default_rng(7)
    gws, ga0, ga1, gu, gd = [rng.normal(size=N).astype(np.float32) for _ in range(5)]
    gimg = rng.normal(size=(N, 3)).astype(np.float32)
    if variant in ("ambient", "sigma"):
        fn = R.composite_rays_train if variant == "ambient" else R.composite_rays_train_sigma
        ws, a0s, dep, img = fn(sg, rg, a0g, dev(dl), dev(rays))
        fo = O.composite_rays_train_forward(variant, sig, rgb, dl, rays, a0)
        outs, grads = [ws, a0s, dep, img], [gws, ga0, gd, gimg]
        assert np.array_equal(host(a0s), fo["amb0_sum"])
        go = O.composite_rays_train_backward(variant, dict(grad_weights_sum=gws, grad_amb0_sum=ga0, grad_image=gimg), sig, rgb, dl, rays,
                                             fo, a0)
    elif variant == "uncertainty":
        ws, a0s, us, dep, img = R.composite_rays_train_uncertainty(sg, rg, a0g, ug, dev(dl), dev(rays))
        fo = O.composite_rays_train_forward(variant, sig, rgb, dl, rays, a0, None, unc)
        outs, grads = [ws, a0s, us, dep, img], [gws, ga0, gu, gd, gimg]
        assert np.array_equal(host(us), fo["unc_sum"])
        go = O.composite_rays_train_backward(variant, dict(grad_weights_sum=gws, grad_amb0_sum=ga0, grad_unc_sum=gu, grad_image=gimg),
                                             sig, rgb, dl, rays, fo, a0, None, unc)
    else:
        ws, a0s, a1s, us, dep, img = R.composite_rays_train_triplane(sg, rg, a0g, a1g, ug, dev(dl), dev(rays))
        fo = O.composite_rays_train_forward(variant, sig, rgb, dl, rays, a0, a1, unc)
        outs, grads = [ws, a0s, a1s, us, dep, img], [gws, ga0, ga1, gu, gd, gimg]
        assert np.array_equal(host(a1s), fo["amb1_sum"]) and np.array_equal(host(us), fo["unc_sum"])
        go = O.composite_rays_train_backward(variant, dict(grad_weights_sum=gws, grad_amb0_sum=ga0, grad_amb1_sum=ga1, grad_unc_sum=gu,
                                                           grad_image=gimg), sig, rgb, dl, rays, fo, a0, a1, unc)
    assert np.array_equal(host(ws), fo["weights_sum"]) and np.array_equal(host(dep), fo["depth"]) and np.array_equal(host(img), fo["image"])
    torch.autograd.backward(outs, [dev(g) for g in grads])
    assert np.array_equal(host(sg.grad), go["grad_sigmas"]) and np.array_equal(host(rg.grad), go["grad_rgbs"])
    assert np.array_equal(host(a0g.grad), go["grad_amb0"])
    if variant == "triplane":
        assert np.array_equal(host(a1g.grad), go["grad_amb1"])
    if variant in ("uncertainty", "triplane"):
        assert np.array_equal(host(ug.grad), go["grad_unc"])


@pytest.mark.parametrize("variant", ["plain", "ambient", "sigma", "uncertainty", "triplane"])
def test_composite_inference_bit_exact(variant):
    from lzzx_nerf_amd import raymarching as R
    rng = np.random.default_rng(11)
    N, n_alive, n_step = 500, 333, 4
    alive = rng.permutation(N)[:n_alive].astype(np.int32)
    M = n_alive * n_step + 128
    sig = rng.uniform(0, 150, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    dl = np.stack([np.full(M, 0.02), np.cumsum(np.full(M, 0.02)) + 2], 1).astype(np.float32)
    for n in range(0, n_alive, 7):  # exhausted rays: zero rows = end-of-ray sentinel (raymarching.cu:2193)
        dl[n * n_step + rng.integers(0, n_step):(n + 1) * n_step] = 0
    a0, a1, unc = [rng.uniform(0, 1, M).astype(np.float32) for _ in range(3)]
    acc = {k: rng.uniform(0, 0.3, N).astype(np.float32) for k in ("ws", "dep", "a0s", "a1s", "us")}
    acc["img"] = rng.uniform(0, 0.3, (N, 3)).astype(np.float32)
    rays_t = rng.uniform(2, 3, N).astype(np.float32)
    o = {k: v.copy() for k, v in acc.items()}
    al_o, rt_o = alive.copy(), rays_t.copy()
    na, aw, hu = O.VARIANTS[variant]
    O.composite_rays(variant, n_alive, n_step, al_o, rt_o, sig, rgb, dl, o["ws"], o["dep"], o["img"], a0 if na > 0 else None,
                     a1 if na > 1 else None, unc if hu else None, o["a0s"] if na > 0 else None, o["a1s"] if na > 1 else None,
                     o["us"] if hu else None, T_thresh=1e-2)
    g = {k: dev(v) for k, v in acc.items()}
    al_g, rt_g = dev(alive), dev(rays_t)
    S, C3, DL = dev(sig), dev(rgb), dev(dl)
    if variant == "plain":
        R.composite_rays(n_alive, n_step, al_g, rt_g, S, C3, DL, g["ws"], g["dep"], g["img"], 1e-2)
    elif variant == "ambient":
        R.composite_rays_ambient(n_alive, n_step, al_g, rt_g, S, C3, DL, dev(a0), g["ws"], g["dep"], g["img"], g["a0s"], 1e-2)
    elif variant == "sigma":
        R.composite_rays_ambient_sigma(n_alive, n_step, al_g, rt_g, S, C3, DL, dev(a0), g["ws"], g["dep"], g["img"], g["a0s"], 1e-2)
    elif variant == "uncertainty":
        R.composite_rays_uncertainty(n_alive, n_step, al_g, rt_g, S, C3, DL, dev(a0), dev(unc), g["ws"], g["dep"], g["img"], g["a0s"],
                                     g["us"], 1e-2)
    else:
        R.composite_rays_triplane(n_alive, n_step, al_g, rt_g, S, C3, DL, dev(a0), dev(a1), dev(unc), g["ws"], g["dep"], g["img"],
                                  g["a0s"], g["a1s"], g["us"], 1e-2)
    assert np.array_equal(host(al_g), al_o) and (al_o < 0).any() and (al_o >= 0).any()
    assert np.array_equal(host(rt_g), rt_o)
    for k in ("ws", "dep", "img") + (("a0s",) if na > 0 else ()) + (("a1s",) if na > 1 else ()) + (("us",) if hu else ()):
        assert np.array_equal(host(g[k]), o[k]), k


# ------------------------------------------------------------------------------------------------
# fused head (MFMA) and the device-resident render loop
# ------------------------------------------------------------------------------------------------
def _head(params, exp_eye=True, precision="f32"):
    from lzzx_nerf_amd.head import FusedTriplaneHead
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    return FusedTriplaneHead(sd, bound=1.0, exp_eye=exp_eye, precision=precision)


@pytest.mark.parametrize("testing", [True, False])
def test_fused_head_bit_exact(params, golden, testing):
    """also the proof that v_mfma_f32_16x16x4_f32 is the k-ordered fma chain the checker assumes"""
    head = _head(params)
    spec = TriplaneSpec(1.0)
    rng = np.random.default_rng(21)
    M = 2000 + 37  # ragged: not a multiple of the 512-sample workgroup tile
    xyz = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    xyz[:4] = [[1, 1, 1], [-1, -1, -1], [0, 0, 0], [1.5, 0.2, 0.1]]
    d = rng.normal(size=(M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    enc_a, eye, ind = golden["net_enc_a"], golden["net_eye"], golden["net_ind"]
    so, ro, ao, eo, uo = head_forward(spec, params, xyz, d, enc_a, ind, eye, testing=testing)
    sg, rg, ag, eg, ug = head.forward(dev(xyz), dev(d), dev(enc_a), dev(ind), dev(eye), testing=testing)
    assert np.array_equal(host(sg), so)
    assert np.array_equal(host(rg), ro)
    assert np.array_equal(host(ag), ao) and np.array_equal(host(eg), eo) and np.array_equal(host(ug), uo)
    # ... and the reference's torch forward on the same inputs (fixture), to torch rounding
    s2, r2, *_ = head.forward(dev(golden["net_xyz"]), dev(golden["net_dirs"]), dev(enc_a), dev(ind), dev(eye), testing=True)
    assert np.allclose(host(s2), golden["net_sigma"], rtol=2e-5, atol=1e-6) and np.allclose(host(r2), golden["net_rgb"], atol=2e-6)


def test_fused_head_count_bound_and_no_eye(params, golden):
    head = _head(params)
    rng = np.random.default_rng(22)
    M = 1500
    xyz, d = rng.uniform(-1, 1, (M, 3)).astype(np.float32), rng.normal(size=(M, 3)).astype(np.float32)
    cnt = torch.tensor([700], dtype=torch.int32, device="cuda")
    out = tuple(torch.full(s, -7.0, device="cuda") for s in ((M,), (M, 3), (M, 1), (M, 1), (M, 1)))
    head.forward(dev(xyz), dev(d), dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]), count_ptr=cnt.data_ptr(), out=out)
    full = head.forward(dev(xyz), dev(d), dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    assert torch.equal(out[0][:700], full[0][:700]) and bool((out[0][700:] == -7).all()) and bool((out[1][700:] == -7).all())
    # exp_eye off: sigma_net sees 68 inputs
    p2 = dict(params)
    p2["sigma_net.net.0.weight"] = np.ascontiguousarray(params["sigma_net.net.0.weight"][:, :68])
    h2 = _head(p2, exp_eye=False)
    spec = TriplaneSpec(1.0)
    so, ro, *_ = head_forward(spec, p2, xyz, d, golden["net_enc_a"], golden["net_ind"], None)
    sg, rg, *_ = h2.forward(dev(xyz), dev(d), dev(golden["net_enc_a"]), dev(golden["net_ind"]), None)
    assert np.array_equal(host(sg), so) and np.array_equal(host(rg), ro)


@pytest.mark.parametrize("boost,use_eye", [(0.0, True), (40.0, True), (0.0, False)])
def test_fused_head_f16_matches_autocast_checker(params, golden, boost, use_eye):
    """precision="f16": the reference's autocast (opt.fp16) rounding sequence on v_mfma_f32_16x16x32_f16.  The order of
    the f32 accumulation inside a half GEMM is the matrix core's (vs numpy's in the checker), so values agree to half
    rounding: the bulk bit-equal, the rest within a few half ulps; sigma = exp(half) within 1 ulp of its half argument."""
    from oracle.head import head_forward_fp16
    p = _scene(params, boost)
    if not use_eye:
        p = dict(p)
        p["sigma_net.net.0.weight"] = np.ascontiguousarray(p["sigma_net.net.0.weight"][:, :68])
    head = _head(p, exp_eye=use_eye, precision="f16")
    rng = np.random.default_rng(31)
    M = 4000 + 7   # ragged last slice
    xyz = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    xyz[:4] = [[1, 1, 1], [-1, -1, -1], [1.0000001, 0, 0], [0, 0, -1.0000001]]
    d = rng.normal(size=(M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    enc_a, ind, eye = golden["net_enc_a"], golden["net_ind"], golden["net_eye"] if use_eye else None
    so, ro, ao, eo, uo = head_forward_fp16(TriplaneSpec(1.0), p, xyz, d, enc_a, ind, eye)
    sg, rg, ag, eg, ug = head.forward(dev(xyz), dev(d), dev(enc_a), dev(ind), None if eye is None else dev(eye))
    sg, rg, ag, ug = host(sg), host(rg), host(ag), host(ug)
    assert np.array_equal(ug, uo)
    assert np.allclose(rg, ro, atol=4e-3) and np.mean(rg == ro) > 0.85, (np.abs(rg - ro).max(), np.mean(rg == ro))
    # sigma = exp in f32 of the half pre-activation: where the half argument agrees the two f32 exps (hardware v_exp_f32 in the kernel, the
    # checker's polynomial) agree to a few ulp; where the argument differs by a half ulp, sigma differs by ~1e-3 relative
    assert np.allclose(sg, so, rtol=2e-2) and np.mean(np.abs(sg / so - 1) < 4e-6) > 0.85, (np.abs(sg / so - 1).max(), np.mean(np.abs(sg / so - 1) < 4e-6))
    assert np.allclose(ag, ao, rtol=5e-3, atol=1e-4)
    if use_eye:
        assert np.allclose(host(eg), eo, atol=2e-3)
    # and it is the half-precision version of the f32 head: close to it, not equal to it
    s32, r32, *_ = _head(p, exp_eye=use_eye).forward(dev(xyz), dev(d), dev(enc_a), dev(ind), None if eye is None else dev(eye))
    assert np.allclose(rg, host(r32), atol=2e-2) and not np.array_equal(rg, host(r32))
    with pytest.raises(RuntimeError, match="inference-only"):
        head.forward(dev(xyz), dev(d), dev(enc_a), dev(ind), None if eye is None else dev(eye), testing=False)


def test_fused_head_f16_matches_reference_autocast_fixture(params, golden):
    """the f16 kernel against the REFERENCE's own forward under torch autocast(float16) (tests/golden/reference_autocast.npz, arrangement
    `h`: enc_a half as encode_audio returns it; produced by tests/golden/make_golden_autocast.py from the imported reference Python).
    rgb / eye attention: the fixture's halves; sigma, ||att||: the fixture ran exp / norm in half (CPU autocast), CUDA autocast and
    the kernel run them in f32 on the same half argument -- compared after rounding to half (tests/test_golden_autocast.py)."""
    ac = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_autocast.npz"), allow_pickle=False)
    head = _head(params, precision="f16")
    enc_a, ind, eye = golden["net_enc_a"], golden["net_ind"], golden["net_eye"]
    sg, rg, ag, eg, ug = (host(t) for t in head.forward(dev(golden["net_xyz"]), dev(golden["net_dirs"]), dev(enc_a), dev(ind), dev(eye)))
    F16 = np.float16

    def ulps(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.abs(a - b) / 2.0 ** (np.floor(np.log2(np.maximum(np.maximum(np.abs(a), np.abs(b)), 2.0 ** -10))) - 10)
    rgb = ac["h_test_rgb"]
    assert ulps(rg, rgb).max() <= 4 and np.mean(rg.astype(F16) == rgb) > 0.85, (ulps(rg, rgb).max(), np.mean(rg.astype(F16) == rgb))
    assert ulps(eg, ac["h_test_amb_eye"]).max() <= 2
    assert ulps(sg.astype(F16), ac["h_test_sigma"]).max() <= 4 and np.mean(sg.astype(F16) == ac["h_test_sigma"]) > 0.85
    assert ulps(ag.astype(F16), ac["h_test_amb_aud"]).max() <= 2
    assert np.allclose(ug, ac["h_test_unc"], atol=1e-7)


def test_render_loop_python_driven_equals_native(params, golden):
    """the per-iteration entries called one by one from Python (debug path) and the native driver lz_loop_run enqueue the
    same launches: same frame, same state, bit for bit"""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    p = _scene(params, 40.0)
    H = W = 96
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = ellipsoid_bitfield()[0]
    enc_a, eye, ind = dev(golden["net_enc_a"]), dev(golden["net_eye"]), dev(golden["net_ind"])
    head = _head(p)
    r1 = TriplaneRenderer(head, dev(bits), bound=1.0)
    a = r1.render(dev(ro), dev(rd), enc_a, ind, eye, max_steps=64, count_samples=True)
    a = {k: v.clone() for k, v in a.items()}
    r2 = TriplaneRenderer(head, dev(bits), bound=1.0)
    r2._head_events = []   # selects the Python-driven loop
    b = r2.render(dev(ro), dev(rd), enc_a, ind, eye, max_steps=64, count_samples=True)
    for k in ("image", "depth", "weights_sum", "ray_counts", "amb_aud_sum", "uncertainty_sum"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["state"][:8], b["state"][:8]) and int(a["state"][72]) == int(b["state"][72]) and len(r2._head_events) >= int(b["state"][6])


def test_render_rgb24_handoff(params, golden):
    """the frame quantised on device equals (pred * 255).astype(np.uint8) of the f32 frame"""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    H = W = 48
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    r = TriplaneRenderer(_head(_scene(params, 40.0)), dev(ellipsoid_bitfield()[0]), bound=1.0)
    out = r.render(dev(ro), dev(rd), dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]), max_steps=48, rgb24=True)
    f32 = host(out["image"])
    assert np.array_equal(host(out["image_rgb24"]), (f32 * 255).astype(np.uint8)) and len(np.unique(host(out["image_rgb24"]))) > 8


def test_render_frame_f16_head(params, golden):
    """whole frame with the f16 head: PSNR against the f32 checker image, marching (independent of the head) unchanged"""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    p = _scene(params, 40.0)
    H = W = 64
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = ellipsoid_bitfield()[0]
    enc_a, eye, ind = golden["net_enc_a"], golden["net_eye"], golden["net_ind"]
    st = {}
    ref = render_inference(TriplaneSpec(1.0), p, ro, rd, bits, enc_a, ind, eye, max_steps=64, stats=st)
    r = TriplaneRenderer(_head(p, precision="f16"), dev(bits), bound=1.0)
    out = r.render(dev(ro), dev(rd), dev(enc_a), dev(ind), dev(eye), max_steps=64, count_samples=True)
    img = host(out["image"])
    mse = float(((img.astype(np.float64) - ref["image"]) ** 2).mean())
    assert mse > 0 and -10 * np.log10(mse) > 50.0, -10 * np.log10(mse)
    cnt = host(out["ray_counts"]).astype(np.int64)
    assert np.mean(cnt == st["samples_per_ray"]) > 0.98 and abs(int(cnt.sum()) - int(st["samples_per_ray"].sum())) < 0.01 * cnt.sum()


def _scene(params, scale_sigma=0.0):
    """denser scene so rays terminate early: bias the sigma row of the last sigma layer"""
    p = dict(params)
    if scale_sigma:
        w = params["sigma_net.net.2.weight"].copy()
        w[0] *= scale_sigma
        p["sigma_net.net.2.weight"] = w
    return p


@pytest.mark.parametrize("scene,max_steps,boost", [("ones", 32, 0.0), ("ellipsoid", 64, 0.0), ("ellipsoid", 64, 40.0)])
def test_render_frame_matches_checker(params, golden, scene, max_steps, boost):
    """whole inference frame: same image, same per-ray sample counts, same iteration schedule as the reference loop"""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    p = _scene(params, boost)
    H = W = 64
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = np.full(128 ** 3 // 8, 255, np.uint8) if scene == "ones" else ellipsoid_bitfield()[0]
    enc_a, eye, ind = golden["net_enc_a"], golden["net_eye"], golden["net_ind"]
    spec = TriplaneSpec(1.0)
    st = {}
    ref = render_inference(spec, p, ro, rd, bits, enc_a, ind, eye, max_steps=max_steps, stats=st)
    r = TriplaneRenderer(_head(p), dev(bits), bound=1.0)
    out = r.render(dev(ro), dev(rd), dev(enc_a), dev(ind), dev(eye), max_steps=max_steps, count_samples=True)
    torch.cuda.synchronize()
    state = host(out["state"])
    assert np.array_equal(host(out["ray_counts"]).astype(np.int64), st["samples_per_ray"])      # per-ray sample counts: bit-exact
    assert state[5] == st["samples_per_ray"].sum() and state[6] == len(st["schedule"]) and state[3] == 1
    assert np.array_equal(host(out["weights_sum"]), ref["weights_sum"])
    assert np.array_equal(host(out["image"]), ref["image"]) and np.array_equal(host(out["depth"]), ref["depth"])
    assert np.array_equal(host(out["amb_aud_sum"]), ref["amb_aud_sum"]) and np.array_equal(host(out["uncertainty_sum"]), ref["uncertainty_sum"])
    if scene == "ellipsoid":
        assert len(set(s for _, s in st["schedule"])) > 1, "scene must exercise n_step > 1 (ray compaction)"
    mse = float(((host(out["image"]).astype(np.float64) - ref["image"]) ** 2).mean())
    assert mse == 0.0  # PSNR vs checker = inf


@pytest.mark.parametrize("scene,boost,factor,cap", [("ones", 0.0, 8, 8), ("ellipsoid", 40.0, 8, 8), ("ellipsoid", 40.0, 4, 16),
                                                    ("ones", 0.0, 4, 4), ("ellipsoid", 40.0, 4, 4)])   # (4, 4) = bench.py's headline
def test_render_frame_fat_schedule(params, golden, scene, boost, factor, cap):
    """sample budget / n_step cap other than the reference's (N, 8): same schedule rule in the checker -> same sample
    counts bit for bit; and the pixels equal those of the reference schedule (rays are independent)"""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    p = _scene(params, boost)
    H = W = 64
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = np.full(128 ** 3 // 8, 255, np.uint8) if scene == "ones" else ellipsoid_bitfield()[0]
    enc_a, eye, ind = golden["net_enc_a"], golden["net_eye"], golden["net_ind"]
    st = {}
    ref = render_inference(TriplaneSpec(1.0), p, ro, rd, bits, enc_a, ind, eye, max_steps=64, stats=st, budget_factor=factor, n_step_cap=cap)
    head = _head(p)
    r = TriplaneRenderer(head, dev(bits), bound=1.0, budget_factor=factor, n_step_cap=cap)
    out = r.render(dev(ro), dev(rd), dev(enc_a), dev(ind), dev(eye), max_steps=64, count_samples=True)
    state = host(out["state"])
    assert np.array_equal(host(out["ray_counts"]).astype(np.int64), st["samples_per_ray"])
    assert state[5] == st["samples_per_ray"].sum() and state[6] == len(st["schedule"]) and state[3] == 1
    assert max(s for _, s in st["schedule"]) > 1
    assert np.array_equal(host(out["image"]), ref["image"]) and np.array_equal(host(out["depth"]), ref["depth"])
    base = TriplaneRenderer(head, dev(bits), bound=1.0).render(dev(ro), dev(rd), dev(enc_a), dev(ind), dev(eye), max_steps=64)
    assert torch.equal(base["image"], out["image"]) and torch.equal(base["weights_sum"], out["weights_sum"])
    assert host(base["state"])[6] > state[6]   # fewer, fatter iterations


def test_dropin_network_path_matches_checker(params, golden):
    """the reference's NeRFNetwork.forward graph (network.py:252-311) built from get_encoder() + torch Linear on the GPU
    operators vs the checker: layouts exact, values to rocBLAS rounding"""
    from lzzx_nerf_amd.encoding import get_encoder
    encs = []
    for n in ("xy", "yz", "xz"):
        e, od = get_encoder("hashgrid", input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=14,
                            desired_resolution=512)
        e = e.cuda()
        e.embeddings.data.copy_(dev(params[f"encoder_{n}.embeddings"]))
        encs.append(e)
    enc_dir, _ = get_encoder("spherical_harmonics")
    x, d = dev(golden["net_xyz"]), dev(golden["net_dirs"])
    enc_x = torch.cat([encs[0](x[:, :2], bound=1), encs[1](x[:, 1:], bound=1), encs[2](x[:, [0, 2]], bound=1)], -1)
    assert np.array_equal(host(enc_x), golden["net_enc_x"])  # features bit-equal to the reference-Python fixture
    W = lambda k: dev(params[k])
    lin = torch.nn.functional.linear
    att = lin(torch.relu(lin(enc_x, W("aud_ch_att_net.net.0.weight"))), W("aud_ch_att_net.net.1.weight"))
    eye_att = torch.sigmoid(lin(torch.relu(lin(enc_x, W("eye_att_net.net.0.weight"))), W("eye_att_net.net.1.weight")))
    h = torch.cat([enc_x, dev(golden["net_enc_a"]) * att, dev(golden["net_eye"]) * eye_att], -1)
    for i in range(3):
        h = lin(h, W(f"sigma_net.net.{i}.weight"))
        if i < 2:
            h = torch.relu(h)
    sigma = torch.exp(h[:, 0])
    hc = torch.cat([enc_dir(d), h[:, 1:], dev(golden["net_ind"]).repeat(x.shape[0], 1)], -1)
    rgb = torch.sigmoid(lin(torch.relu(lin(hc, W("color_net.net.0.weight"))), W("color_net.net.1.weight"))) * 1.002 - 0.001
    assert np.allclose(host(sigma), golden["net_sigma"], rtol=1e-4, atol=1e-5)
    assert np.allclose(host(rgb), golden["net_rgb"], atol=1e-5)


def test_train_forward_matches_checker(params, golden):
    """training branch of run_cuda (renderer.py:279-304) through the operator API + fused head"""
    from lzzx_nerf_amd import raymarching as R
    H = W = 48
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    bits = ellipsoid_bitfield()[0]
    enc_a, eye, ind = golden["net_enc_a"], golden["net_eye"], golden["net_ind"]
    spec = TriplaneSpec(1.0)
    ref = render_train_forward(spec, params, ro, rd, bits, enc_a, ind, eye, max_steps=48, force_all_rays=True)
    aabb = dev(np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32))
    nears, fars = R.near_far_from_aabb(dev(ro), dev(rd), aabb, 0.05)
    ctr = torch.zeros(2, dtype=torch.int32, device="cuda")
    xyzs, dirs, deltas, rays = R.march_rays_train(dev(ro), dev(rd), 1.0, dev(bits), 1, 128, nears, fars, ctr, -1, False, 128, True, 1 / 256, 48)
    sig, rgb, aa, ae, unc = _head(params).forward(xyzs.contiguous(), dirs.contiguous(), dev(enc_a), dev(ind), dev(eye), testing=False)
    ws, a0s, a1s, us, dep, img = R.composite_rays_train_triplane(sig, rgb, aa.abs().sum(-1), ae.abs().sum(-1), unc, deltas, rays)
    c = ref["comp"]
    assert np.array_equal(host(rays), ref["rays"]) and np.array_equal(host(sig), ref["sigmas"])
    assert np.array_equal(host(ws), c["weights_sum"]) and np.array_equal(host(img), c["image"]) and np.array_equal(host(us), c["unc_sum"])
    assert np.array_equal(host(a0s), c["amb0_sum"]) and np.array_equal(host(a1s), c["amb1_sum"]) and np.array_equal(host(dep), c["depth"])


# ------------------------------------------------------------------------------------------------
# BASELINE-size properties (the checker is too slow there; use size-independent invariants)
# ------------------------------------------------------------------------------------------------
def test_full_size_frame_properties(params, golden):
    from lzzx_nerf_amd.renderer import TriplaneRenderer, get_rays as gpu_get_rays
    H = W = 512
    pose, intr = synthetic_camera(H, W)
    ro, rd = gpu_get_rays(dev(pose), intr, H, W)
    bits, _ = ellipsoid_bitfield()
    head = _head(_scene(params, 40.0))
    r = TriplaneRenderer(head, dev(bits), bound=1.0)
    enc_a, eye, ind = dev(golden["net_enc_a"]), dev(golden["net_eye"]), dev(golden["net_ind"])
    full = r.render(ro, rd, enc_a, ind, eye, max_steps=192, count_samples=True)
    img, ws, cnt = full["image"].clone(), full["weights_sum"].clone(), full["ray_counts"].clone()
    st = host(full["state"])
    assert st[3] == 1 and st[5] == int(cnt.sum()) and 0 < st[5] < H * W * 192
    assert bool(((img >= 0) & (img <= 1)).all()) and bool((ws >= 0).all()) and float(ws.max()) <= 1 + 1e-5
    # rays are independent: rendering any subset (different N => different n_step schedule) gives the same pixels
    sel = torch.randperm(H * W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))[:50000].sort().values
    sub = r.render(ro[sel].contiguous(), rd[sel].contiguous(), enc_a, ind, eye, max_steps=192, count_samples=True)
    assert torch.equal(sub["image"], img[sel]) and torch.equal(sub["ray_counts"], cnt[sel])
    # rays that miss the ellipsoid see pure background
    miss = cnt == 0
    assert bool(miss.any()) and bool((img[miss] == 1).all())
    # determinism
    again = r.render(ro, rd, enc_a, ind, eye, max_steps=192)
    assert torch.equal(again["image"], img)


@pytest.mark.parametrize("bound,n_rays,max_steps", [(2.0, 1000, 40), (1.0, 37, 5), (4.0, 513, 24)])
def test_render_odd_sizes_and_cascades(params, golden, bound, n_rays, max_steps):
    """ray counts that are not multiples of a workgroup, very few steps, bound > 1 (cascade 2 / 3: mip levels in the march, larger
    tables in the head): whole frame against the checker, bit for bit"""
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    rng = np.random.default_rng(int(bound * 7) + n_rays)
    spec = TriplaneSpec(bound)
    p = dict(_scene(params, 30.0))
    if bound != 1.0:
        for n in ("xy", "yz", "xz"):
            p[f"encoder_{n}.embeddings"] = rng.uniform(-1, 1, (spec.n_params, 1)).astype(np.float32)
            p[f"encoder_{n}.offsets"] = spec.offsets.astype(np.int32)
    cascade = 1 + int(np.ceil(np.log2(bound)))
    # random occupancy, 30 % of the cells of every cascade
    grid = (rng.uniform(size=(cascade, 128 ** 3)) < 0.3).astype(np.float32)
    bits = O.packbits(grid, 0.5)
    pose, intr = synthetic_camera(64, 64)
    ro, rd = get_rays(pose, intr, 64, 64)
    sel = rng.choice(64 * 64, n_rays, replace=False)
    ro, rd = np.ascontiguousarray(ro[sel]) * np.float32(bound), np.ascontiguousarray(rd[sel])
    enc_a, eye, ind = golden["net_enc_a"], golden["net_eye"], golden["net_ind"]
    st = {}
    ref = render_inference(spec, p, ro, rd, bits, enc_a, ind, eye, cascade=cascade, max_steps=max_steps, stats=st)
    head = FusedTriplaneHead({k: torch.from_numpy(np.asarray(v)) for k, v in p.items()}, bound=bound)
    r = TriplaneRenderer(head, dev(bits), bound=bound)
    out = r.render(dev(ro), dev(rd), dev(enc_a), dev(ind), dev(eye), max_steps=max_steps, count_samples=True)
    assert np.array_equal(host(out["ray_counts"]).astype(np.int64), st["samples_per_ray"])
    assert np.array_equal(host(out["image"]), ref["image"]) and np.array_equal(host(out["depth"]), ref["depth"])
    state = host(out["state"])
    assert state[3] == 1 and state[5] == st["samples_per_ray"].sum() and state[6] == len(st["schedule"]) and st["samples_per_ray"].sum() > 0


def test_render_more_rays_than_one_pass(params, golden, monkeypatch):
    """batches above the device loop's 2^20-ray limit are rendered in passes; pixels equal the single-pass render"""
    import lzzx_nerf_amd.renderer as RR
    H = W = 96
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    r = RR.TriplaneRenderer(_head(_scene(params, 40.0)), dev(ellipsoid_bitfield()[0]), bound=1.0)
    args = (dev(ro), dev(rd), dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    one = {k: v.clone() for k, v in r.render(*args, max_steps=48, count_samples=True).items()}
    monkeypatch.setattr(RR, "MAX_RAYS_PER_PASS", 4000)   # 9216 rays -> 3 passes, ragged last one
    many = r.render(*args, max_steps=48, count_samples=True)
    for k in ("image", "depth", "weights_sum", "ray_counts"):
        assert torch.equal(one[k], many[k]), k
    assert int(many["state"][5]) == int(one["state"][5])


def test_full_size_grid_linearity_and_layouts():
    """B = 2^22 samples, cfg2 table (49 MB): f(a e1 + b e2) = a f(e1) + b f(e2) up to rounding; both output layouts agree"""
    from lzzx_nerf_amd._util import call, ptr, stream
    from lzzx_nerf_amd.gridencoder import GridEncoder, grid_encode
    enc = GridEncoder(desired_resolution=2048).cuda()  # get_encoder('hashgrid') defaults: D=3, L=16, C=2, H=16, T=2^19, res 2048
    B = 1 << 22
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(B, 3, device="cuda", generator=g)
    e1 = torch.rand(enc.embeddings.shape, device="cuda", generator=g) - 0.5
    e2 = torch.rand(enc.embeddings.shape, device="cuda", generator=g) - 0.5
    f = lambda e: grid_encode(x, e, enc.offsets, enc.per_level_scale, 16, False, 0, False)
    f1, f2, f12 = f(e1), f(e2), f(2 * e1 - 0.5 * e2)
    assert float((f12 - (2 * f1 - 0.5 * f2)).abs().max()) < 1e-5
    lm = torch.empty(16, B, 2, device="cuda")
    call("lz_grid_encode_forward", ptr(x), ptr(e1), ptr(enc.offsets), ptr(lm), B, 3, 2, 16, float(np.float32(np.log2(enc.per_level_scale))),
         16, None, 0, 0, 0, 0, stream())
    assert torch.equal(lm.permute(1, 0, 2).reshape(B, 32), f1)


def test_composite_reference_named_entry_points():
    """the C ABI exports the reference's 13 compositing functions under their own names and argument order (raymarching.h:16-38);
    they must give what the descriptor-driven entries behind the Python operators give"""
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd._util import call, ptr, stream
    g = torch.Generator(device="cuda").manual_seed(7)
    N, S = 300, 6
    M = N * S
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g)
    sig, rgb, a0, a1, un = rnd(M) * 3, rnd(M, 3), rnd(M), rnd(M), rnd(M)
    deltas = torch.stack([torch.full((M,), 0.02, device="cuda"), torch.arange(M, device="cuda").float() * 0.02 + 1], 1).contiguous()
    rays = torch.stack([torch.arange(N, device="cuda"), torch.arange(N, device="cuda") * S, torch.full((N,), S, device="cuda")], 1).int().contiguous()
    z = lambda *s: torch.zeros(*s, device="cuda")
    # train forward / backward, triplane variant
    ws, aas, aes, us, dep, img = R.composite_rays_train_triplane(sig, rgb, a0, a1, un, deltas, rays, 1e-4)
    o = [z(N), z(N), z(N), z(N), z(N), z(N, 3)]
    call("lz_composite_rays_train_triplane_forward", ptr(sig), ptr(rgb), ptr(a0), ptr(a1), ptr(un), ptr(deltas), ptr(rays), M, N, 1e-4,
         *[ptr(t) for t in o], stream())
    for a, b in zip((ws, aas, aes, us, dep, img), o):
        assert torch.equal(a, b)
    gws, gaa, gae, gu, gim = rnd(N), rnd(N), rnd(N), rnd(N), rnd(N, 3)
    ref = R._composite_train_bwd((2, 0, 1), gws, gaa, gae, gu, gim, sig, rgb, a0, a1, un, deltas, rays, ws, aas, us, img, 1e-4)
    go = [z(M), z(M, 3), z(M), z(M), z(M)]
    call("lz_composite_rays_train_triplane_backward", ptr(gws), ptr(gaa), ptr(gae), ptr(gu), ptr(gim), ptr(sig), ptr(rgb), ptr(a0), ptr(a1), ptr(un),
         ptr(deltas), ptr(rays), ptr(ws), ptr(aas), ptr(aes), ptr(us), ptr(img), M, N, 1e-4, *[ptr(t) for t in go], stream())
    for a, b in zip(ref, go):
        assert torch.equal(a, b)
    # sigma-weighted ambient variant, forward
    ws2, as2, dep2, img2 = R.composite_rays_train_sigma(sig, rgb, a0, deltas, rays, 1e-4)
    o2 = [z(N), z(N), z(N), z(N, 3)]
    call("lz_composite_rays_train_sigma_forward", ptr(sig), ptr(rgb), ptr(a0), ptr(deltas), ptr(rays), M, N, 1e-4, *[ptr(t) for t in o2], stream())
    for a, b in zip((ws2, as2, dep2, img2), o2):
        assert torch.equal(a, b)
    # inference: triplane and plain
    n_alive, n_step = N, S
    mk = lambda: (torch.arange(N, device="cuda").int(), torch.ones(N, device="cuda"), z(N), z(N), z(N, 3), z(N), z(N), z(N))
    al, rt, w_, d_, im_, s0, s1, s2 = mk()
    R.composite_rays_triplane(n_alive, n_step, al, rt, sig, rgb, deltas, a0, a1, un, w_, d_, im_, s0, s1, s2, 1e-2)
    al2, rt2, w2, d2, im2, t0, t1, t2 = mk()
    call("lz_composite_rays_triplane", n_alive, n_step, 1e-2, ptr(al2), ptr(rt2), ptr(sig), ptr(rgb), ptr(deltas), ptr(a0), ptr(a1), ptr(un), ptr(w2),
         ptr(d2), ptr(im2), ptr(t0), ptr(t1), ptr(t2), stream())
    for a, b in zip((al, rt, w_, d_, im_, s0, s1, s2), (al2, rt2, w2, d2, im2, t0, t1, t2)):
        assert torch.equal(a, b)
    al, rt, w_, d_, im_, *_ = mk()
    R.composite_rays(n_alive, n_step, al, rt, sig, rgb, deltas, w_, d_, im_, 1e-2)
    al2, rt2, w2, d2, im2, *_ = mk()
    call("lz_composite_rays", n_alive, n_step, 1e-2, ptr(al2), ptr(rt2), ptr(sig), ptr(rgb), ptr(deltas), ptr(w2), ptr(d2), ptr(im2), stream())
    for a, b in zip((al, rt, w_, d_, im_), (al2, rt2, w2, d2, im2)):
        assert torch.equal(a, b)


def test_grid_backward_level_resident_on_runs_of_equal_cells():
    """march-ordered samples of rays normal to a plane fall into the SAME cell in long runs; lz_k_grid_backward_lds_fx deals the lanes of a
    wave-instruction rows that lie a whole segment of the chunk apart (lane-major segments; rounds 3-4: a per-workgroup vote and a 4-row
    spread) so that the LDS atomics do not all hit one address.  The fixed-point sums are exact integers, so the result must not depend on
    the order: runs of 1 .. 200 equal positions (plus jitter far
    below a cell of the coarse levels) against the checker and against the same call on a shuffled copy of the samples."""
    from lzzx_nerf_amd.gridencoder import GridEncoder
    from lzzx_nerf_amd._util import call, ptr, stream
    D, L, C, H, T, res = 2, 12, 1, 64, 14, 512
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=T, desired_resolution=res).cuda()
    rng = np.random.default_rng(5)
    B = 60000
    lens = rng.integers(1, 200, size=2000)
    base = rng.uniform(0.05, 0.95, (len(lens), D)).astype(np.float32)
    x = np.repeat(base, lens, axis=0)[:B]
    x = (x + rng.uniform(-2e-4, 2e-4, x.shape)).astype(np.float32)
    assert x.shape[0] == B
    g = rng.normal(size=(B, L * C)).astype(np.float32)
    off = host(enc.offsets)
    ge, _ = O.grid_encode_backward(g, x, tuple(enc.embeddings.shape), off, enc.per_level_scale, H, None, 0)
    S = float(np.float32(np.log2(enc.per_level_scale)))
    outs = []
    perm = rng.permutation(B)
    for xs, gs in ((x, g), (x[perm], g[perm])):
        gemb = torch.zeros_like(enc.embeddings.data)
        gl = dev(np.ascontiguousarray(gs.reshape(B, L, C).transpose(1, 0, 2)))      # level-major, as the fused training head hands it over
        call("lz_grid_encode_backward", ptr(gl), ptr(dev(xs)), ptr(enc.embeddings.data), ptr(enc.offsets), ptr(gemb), B, D, C, L, S, H,
             None, None, 0, 0, 0, 3, stream())
        outs.append(host(gemb))
    scale = np.abs(ge).max()
    assert np.max(np.abs(outs[0] - ge)) < 1e-4 * scale
    # same terms, exact accumulation inside a workgroup; only the f32 flush of the per-chunk sums can differ in its last bits
    assert np.max(np.abs(outs[0] - outs[1])) < 2e-6 * scale
