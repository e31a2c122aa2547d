"""Known-answer tests that pin the CPU checker itself (the reference ships no tests).
Every expected value here comes from an independent closed form, not from the code under test."""
import numpy as np
import pytest
from scipy import special

from oracle import oracle as O


def test_detmath_against_libm():
    x = np.linspace(-80, 80, 200001).astype(np.float32)
    e = O.unary("exp", x)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 1.5e-7
    xs = np.linspace(-200, 200, 200001).astype(np.float32)
    assert np.max(np.abs(O.unary("sin", xs) - np.sin(xs.astype(np.float64)))) < 1.5e-7
    xl = np.exp(np.linspace(-80, 80, 100001)).astype(np.float32)
    assert np.max(np.abs(O.unary("log", xl) - np.log(xl.astype(np.float64)))) < 4e-6
    s = O.unary("sigmoid", x)
    assert np.max(np.abs(s - 1 / (1 + np.exp(-x.astype(np.float64))))) < 2e-7
    assert O.unary("softplus", np.zeros(1, np.float32))[0] == pytest.approx(np.log(2), abs=1e-7)


def test_morton_roundtrip_and_bit_interleave():
    c = np.arange(128, dtype=np.int32)
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    idx = O.morton3D(coords)
    assert idx.min() == 0 and idx.max() == 128 ** 3 - 1 and len(np.unique(idx)) == 128 ** 3
    assert np.array_equal(O.morton3D_invert(idx), coords)
    # independent bit interleave
    def spread(v):
        r = np.zeros_like(v, dtype=np.int64)
        for b in range(10):
            r |= ((v.astype(np.int64) >> b) & 1) << (3 * b)
        return r
    exp = spread(coords[:, 0]) | (spread(coords[:, 1]) << 1) | (spread(coords[:, 2]) << 2)
    assert np.array_equal(idx.astype(np.int64), exp)


def test_packbits_matches_numpy_little_endian():
    rng = np.random.default_rng(0)
    g = rng.uniform(0, 1, (2, 4096)).astype(np.float32)
    bits = O.packbits(g, 0.5)
    assert np.array_equal(bits, np.packbits((g > 0.5).reshape(-1), bitorder="little"))


def test_dilation_is_six_neighbour_max():
    H = 8
    rng = np.random.default_rng(1)
    dense = rng.uniform(0, 1, (H, H, H)).astype(np.float32)
    c = np.arange(H, dtype=np.int32)
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    idx = O.morton3D(coords)
    grid = np.zeros((1, H ** 3), np.float32)
    grid[0, idx] = dense.ravel()
    out = O.morton3D_dilation(grid)[0, idx].reshape(H, H, H)
    pad = np.pad(dense, 1, constant_values=-np.inf)
    exp = np.max(np.stack([pad[1:-1, 1:-1, 1:-1], pad[2:, 1:-1, 1:-1], pad[:-2, 1:-1, 1:-1], pad[1:-1, 2:, 1:-1],
                           pad[1:-1, :-2, 1:-1], pad[1:-1, 1:-1, 2:], pad[1:-1, 1:-1, :-2]]), 0)
    assert np.array_equal(out, exp)


def test_near_far_slab():
    rng = np.random.default_rng(2)
    o = np.tile(np.array([[0, 0, -3.35]], np.float32), (1000, 1))
    d = rng.normal(size=(1000, 3)).astype(np.float32) * np.array([0.3, 0.3, 1], np.float32)
    d[:, 2] = np.abs(d[:, 2]) + 0.5
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    n, f = O.near_far_from_aabb(o, d, aabb, 0.05)
    o64, d64 = o.astype(np.float64), d.astype(np.float64)
    t1, t2 = (aabb[:3] - o64) / d64, (aabb[3:] - o64) / d64
    tn, tf = np.minimum(t1, t2).max(1), np.maximum(t1, t2).min(1)
    hit = tn <= tf
    assert np.all(n[~hit] == np.finfo(np.float32).max) and np.all(f[~hit] == np.finfo(np.float32).max)
    assert np.allclose(n[hit], np.maximum(tn[hit], 0.05), rtol=1e-5) and np.allclose(f[hit], tf[hit], rtol=1e-5)
    assert hit.sum() > 100 and (~hit).sum() > 10


def _real_sh(l, m, theta, phi):
    """real SH with Condon-Shortley phase from scipy's complex Y_l^m (theta polar, phi azimuth)"""
    if hasattr(special, "sph_harm_y"):
        Y = lambda mm: special.sph_harm_y(l, mm, theta, phi)
    else:
        Y = lambda mm: special.sph_harm(mm, l, phi, theta)
    if m == 0:
        return Y(0).real
    if m > 0:
        return np.sqrt(2) * (-1) ** m * Y(m).real * (-1) ** m  # scipy's Y already carries the CS phase
    return np.sqrt(2) * Y(-m).imag


@pytest.mark.parametrize("degree", [1, 4, 8])
def test_sh_against_scipy(degree):
    rng = np.random.default_rng(3)
    d = rng.normal(size=(500, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out, _ = O.sh_encode_forward(d.astype(np.float32), degree)
    theta, phi = np.arccos(np.clip(d[:, 2], -1, 1)), np.arctan2(d[:, 1], d[:, 0])
    for l in range(degree):
        for m in range(-l, l + 1):
            exp = _real_sh(l, m, theta, phi)
            assert np.max(np.abs(out[:, l * l + l + m] - exp)) < 3e-5, (l, m)


def test_sh_axis_values_and_jacobian():
    # closed forms quoted in the reference's comments (shencoder.cu:50-68)
    out, _ = O.sh_encode_forward(np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], np.float32), 4)
    assert out[0, 0] == pytest.approx(0.5 / np.sqrt(np.pi), rel=1e-6)
    assert out[0, 2] == pytest.approx(np.sqrt(3) / (2 * np.sqrt(np.pi)), rel=1e-6)      # +z
    assert out[1, 3] == pytest.approx(-np.sqrt(3) / (2 * np.sqrt(np.pi)), rel=1e-6)     # -x
    assert out[2, 1] == pytest.approx(-np.sqrt(3) / (2 * np.sqrt(np.pi)), rel=1e-6)     # -y
    assert out[0, 6] == pytest.approx(np.sqrt(5) * 2 / (4 * np.sqrt(np.pi)), rel=1e-6)  # sqrt5 (3z^2-1)/(4 sqrt pi)
    assert out[0, 12] == pytest.approx(np.sqrt(7) * 2 / (4 * np.sqrt(np.pi)), rel=1e-6)
    # Jacobian vs central differences of the polynomial (non-unit inputs on purpose)
    rng = np.random.default_rng(4)
    x = rng.uniform(-0.9, 0.9, (50, 3)).astype(np.float32)
    _, J = O.sh_encode_forward(x, 6, True)
    J = J.reshape(50, 3, 36)
    h = 1e-2
    for k in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, k] += h
        xm[:, k] -= h
        fd = (O.sh_encode_forward(xp, 6)[0].astype(np.float64) - O.sh_encode_forward(xm, 6)[0]) / (xp[:, k] - xm[:, k])[:, None]
        assert np.max(np.abs(J[:, k] - fd)) < 5e-2 * max(1.0, np.abs(fd).max()) * h * 10
    g = rng.normal(size=(50, 36)).astype(np.float32)
    gi = O.sh_encode_backward(g, J.reshape(50, -1), 6)
    assert np.allclose(gi, np.einsum("bc,bkc->bk", g.astype(np.float64), J.astype(np.float64)), atol=1e-4)


def test_freq_against_numpy():
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (300, 2)).astype(np.float32)
    out = O.freq_encode_forward(x, 8)
    assert out.shape == (300, 34)
    assert np.array_equal(out[:, :2], x)
    for f in range(8):
        assert np.allclose(out[:, 2 + 4 * f: 4 + 4 * f], np.sin(2.0 ** f * x.astype(np.float64)), atol=2e-6)
        # cos is sin(2^f x + float(pi/2)) (freqencoder.cu:54-56): the f32 add costs up to ulp(2^f)/2
        assert np.allclose(out[:, 4 + 4 * f: 6 + 4 * f], np.cos(2.0 ** f * x.astype(np.float64)), atol=2e-6 + 2.0 ** f * 7e-8)
    g = rng.normal(size=out.shape).astype(np.float32)
    gi = O.freq_encode_backward(g, out, 2, 8)
    x64 = x.astype(np.float64)
    exp = g[:, :2].astype(np.float64)
    for f in range(8):
        exp += 2.0 ** f * (g[:, 2 + 4 * f: 4 + 4 * f] * np.cos(2.0 ** f * x64) - g[:, 4 + 4 * f: 6 + 4 * f] * np.sin(2.0 ** f * x64))
    assert np.allclose(gi, exp, atol=2e-3)


def test_grid_dense_level_is_bilinear_interpolation():
    # a single dense 2-D level: value at x must equal bilinear interpolation of the (res+1)^2 lattice at x*scale+0.5
    H, D = 16, 2
    off = O.grid_offsets(D, 1, 2.0, H, 19)
    assert off[1] == int(np.ceil((H + 1) ** 2 / 8) * 8)
    rng = np.random.default_rng(6)
    emb = rng.normal(size=(off[1], 1)).astype(np.float32)
    x = rng.uniform(0, 1, (400, 2)).astype(np.float32)
    x[0] = [0, 0]
    x[1] = [1, 1]
    out, dydx = O.grid_encode_forward(x, emb, off, 2.0, H, True)
    scale = H - 1.0
    pos = x.astype(np.float64) * scale + 0.5
    p0 = np.floor(pos).astype(int)
    fr = pos - p0
    lat = lambda ix, iy: emb[(ix + iy * (H + 1)) % off[1], 0].astype(np.float64)
    exp = ((1 - fr[:, 0]) * (1 - fr[:, 1]) * lat(p0[:, 0], p0[:, 1]) + fr[:, 0] * (1 - fr[:, 1]) * lat(p0[:, 0] + 1, p0[:, 1]) +
           (1 - fr[:, 0]) * fr[:, 1] * lat(p0[:, 0], p0[:, 1] + 1) + fr[:, 0] * fr[:, 1] * lat(p0[:, 0] + 1, p0[:, 1] + 1))
    assert np.allclose(out[:, 0], exp, atol=2e-5)
    # d/dx0 = scale * ((1-fy) (v10 - v00) + fy (v11 - v01))
    ex0 = scale * ((1 - fr[:, 1]) * (lat(p0[:, 0] + 1, p0[:, 1]) - lat(p0[:, 0], p0[:, 1])) +
                   fr[:, 1] * (lat(p0[:, 0] + 1, p0[:, 1] + 1) - lat(p0[:, 0], p0[:, 1] + 1)))
    assert np.allclose(dydx.reshape(400, 1, 2, 1)[:, 0, 0, 0], ex0, atol=2e-3)
    # out of range -> zeros (gridencoder.cu:98-122); exactly 1.0 is in range
    o2, _ = O.grid_encode_forward(np.array([[1.0001, 0.5], [-1e-6, 0.5], [1.0, 1.0]], np.float32), emb, off, 2.0, H)
    assert o2[0, 0] == 0 and o2[1, 0] == 0 and o2[2, 0] != 0


def test_grid_hash_index_formula_and_backward_is_transpose():
    D, L, C, H = 3, 4, 2, 16
    pls = 2.0
    off = O.grid_offsets(D, L, pls, H, 12)  # small table -> upper levels hashed
    rng = np.random.default_rng(7)
    x = rng.uniform(0, 1, (200, 3)).astype(np.float32)
    idx = O.grid_corner_indices(x, off, C, pls, H)
    sc, res = O.grid_level_params(L, np.float32(np.log2(pls)), H)
    primes = np.array([1, 2654435761, 805459861], dtype=np.uint64)
    for l in range(L):
        hs = int(off[l + 1] - off[l])
        pg = np.floor(x.astype(np.float32) * sc[l] + np.float32(0.5)).astype(np.uint64)
        for c in range(8):
            pl = pg + np.array([(c >> d) & 1 for d in range(3)], dtype=np.uint64)
            stride, index, hashed = 1, np.zeros(200, dtype=np.uint64), False
            for d in range(3):
                if stride <= hs:
                    index = (index + pl[:, d] * np.uint64(stride)) & np.uint64(0xFFFFFFFF)
                    stride *= int(res[l]) + 1
            if stride > hs:
                index = ((pl[:, 0] * primes[0]) & np.uint64(0xFFFFFFFF)) ^ ((pl[:, 1] * primes[1]) & np.uint64(0xFFFFFFFF)) ^ \
                        ((pl[:, 2] * primes[2]) & np.uint64(0xFFFFFFFF))
            exp = (int(off[l]) + (index % np.uint64(hs)).astype(np.int64)) * C
            assert np.array_equal(idx[l, :, c].astype(np.int64), exp), (l, c)
    # backward == transpose of forward: <grad, f(emb)> == <grad_emb, emb> (f is linear in emb)
    emb = rng.normal(size=(off[-1], C)).astype(np.float32)
    out, _ = O.grid_encode_forward(x, emb, off, pls, H)
    g = rng.normal(size=out.shape).astype(np.float32)
    ge, _ = O.grid_encode_backward(g, x, emb.shape, off, pls, H)
    assert np.sum(g.astype(np.float64) * out) == pytest.approx(np.sum(ge.astype(np.float64) * emb), rel=1e-4)


def test_grid_half_tables_round_like_at_half():
    D, L, C, H = 2, 3, 2, 8
    off = O.grid_offsets(D, L, 2.0, H, 10)
    rng = np.random.default_rng(8)
    emb = rng.normal(size=(off[-1], C)).astype(np.float16)
    x = rng.uniform(0, 1, (100, 2)).astype(np.float32)
    oh, _ = O.grid_encode_forward(x, emb, off, 2.0, H)
    of, _ = O.grid_encode_forward(x, emb.astype(np.float32), off, 2.0, H)
    assert oh.dtype == np.float16
    assert np.max(np.abs(oh.astype(np.float32) - of)) < 1e-2 and np.any(oh.astype(np.float32) != of)


def test_grid_align_corners_dense_levels_are_grid_sample_align_corners():
    """align_corners=True (encoding.py:9,25,29; gridencoder.cu:62,135): scale = H s^l - 1 with NO half-cell offset, side = resolution
    (not resolution + 1): lattice node i sits at x = i / (res - 1), i.e. torch's grid_sample(align_corners=True) over a res x res image"""
    import torch
    import torch.nn.functional as F
    D, L, H, pls = 2, 3, 8, 2.0
    off = O.grid_offsets(D, L, pls, H, 19, align_corners=True)
    assert list(np.diff(off)) == [64, 256, 1024]                      # side^2 = (H 2^l)^2, already multiples of 8
    assert list(np.diff(O.grid_offsets(D, L, pls, H, 19))) == [88, 296, 1096]   # (side + 1)^2 rounded up to 8 without it
    rng = np.random.default_rng(16)
    emb = rng.normal(size=(off[-1], 1)).astype(np.float32)
    x = rng.uniform(0, 1, (500, 2)).astype(np.float32)
    x[:4] = [[0, 0], [1, 1], [1, 0], [0.5, 0.5]]
    out, dydx = O.grid_encode_forward(x, emb, off, pls, H, True, 0, True)
    g = torch.from_numpy(x.astype(np.float64) * 2 - 1).view(1, -1, 1, 2)
    for l in range(L):
        side = H * 2 ** l
        img = torch.from_numpy(emb[off[l]:off[l + 1], 0].astype(np.float64)).view(1, 1, side, side)   # index = x + y * side
        ref = F.grid_sample(img, g, mode="bilinear", padding_mode="border", align_corners=True).view(-1).numpy()
        assert np.allclose(out[:, l], ref, atol=3e-6), l
    assert out[0, 0] == emb[0, 0] and out[1, 0] == emb[63, 0] and out[2, 0] == emb[7, 0]      # nodes are hit exactly at 0 and 1
    # indices: corner (0, 0) of x = (1, 1) is the last node; x + 1 / y + 1 neighbours carry weight 0 and wrap modulo the level size
    idx = O.grid_corner_indices(x[:2], off, 1, pls, H, 0, True)
    assert idx[0, 0, 0] == 0 and idx[0, 1, 0] == 63 and idx[1, 1, 0] == off[1] + 255
    # dy/dx0 at the centre of a cell = scale * (v10 - v00 ...): finite differences of the interpolant
    eps = 1e-3
    xm = x[4:].copy()
    xm = xm[(np.abs((xm * (H - 1)) % 1 - 0.5) < 0.3).all(1)]           # stay inside level 0's cell for the central difference
    op, _ = O.grid_encode_forward(xm + np.float32([eps, 0]), emb, off, pls, H, False, 0, True)
    om, _ = O.grid_encode_forward(xm - np.float32([eps, 0]), emb, off, pls, H, False, 0, True)
    _, dd = O.grid_encode_forward(xm, emb, off, pls, H, True, 0, True)
    assert np.allclose(dd.reshape(-1, L, 2, 1)[:, 0, 0, 0], (op[:, 0] - om[:, 0]) / (2 * eps), atol=2e-2)
    # differs from the default convention on the same table
    o2, _ = O.grid_encode_forward(x, emb, off, pls, H, False, 0, False)
    assert not np.allclose(o2[4:], out[4:], atol=1e-3)


def test_grid_half_backward_restatement():
    """half tables (gridencoder.cu:296-311): every term is half(w * g), every add is a half + half -> half into the table"""
    D, L, C, H, pls = 2, 3, 2, 8, 2.0
    off = O.grid_offsets(D, L, pls, H, 9)
    rng = np.random.default_rng(17)
    B = 60
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = [1.0000001, 0.5]                                            # out of range: contributes nothing
    g = rng.normal(size=(B, L * C)).astype(np.float16)
    r = O.grid_encode_backward_f16(g, x, (off[-1], C), off, pls, H)
    idx = O.grid_corner_indices(x, off, C, pls, H)                      # [L, B, 4] element index of channel 0 (offset included)
    sc, _ = O.grid_level_params(L, np.float32(np.log2(pls)), H)
    ge = np.zeros((off[-1] * C,), np.float16)
    exact = np.zeros((off[-1] * C,), np.float64)
    cnt = np.zeros((off[-1] * C,), np.int32)
    for l in range(L):
        for b in range(B):
            if idx[l, b, 0] < 0:
                continue
            pos = (x[b].astype(np.float64) * float(sc[l]) + 0.5).astype(np.float32)      # one rounding: the kernel's fma
            fr = pos - np.floor(pos)
            for c in range(4):
                w = np.float32(1)
                for d in range(2):
                    w = np.float32(w * (fr[d] if (c >> d) & 1 else np.float32(1) - fr[d]))
                for ch in range(C):
                    term = np.float16(w * np.float32(g[b, l * C + ch]))
                    e = idx[l, b, c] + ch
                    ge[e] = np.float16(np.float32(ge[e]) + np.float32(term))
                    exact[e] += float(term)
                    cnt[e] += 1
    assert np.array_equal(r["grad_embeddings"].reshape(-1).view(np.uint16), ge.view(np.uint16))
    assert np.array_equal(r["terms"].reshape(-1), cnt) and np.allclose(r["exact"].reshape(-1), exact, rtol=0, atol=1e-12)
    assert cnt.max() >= 3 and (cnt == 0).any() and r["terms"].sum() == (B - 1) * L * 4 * C
    # an entry with one or two terms does not depend on the order (half addition commutes); any entry stays within its rounding budget
    few = cnt <= 2
    assert np.array_equal(ge[few].astype(np.float64), exact[few].astype(np.float16).astype(np.float64))
    assert np.all(np.abs(ge.astype(np.float64) - exact) <= np.maximum(cnt - 1, 0) * 2.0 ** -11 * np.maximum(r["absum"].reshape(-1), 2.0 ** -14) + 1e-12)
    # grad_inputs with half dy_dx: Half += Half * Half, sequential over (level, channel)
    emb = rng.normal(size=(off[-1], C)).astype(np.float16)
    _, dd = O.grid_encode_forward(x, emb, off, pls, H, True)
    r2 = O.grid_encode_backward_f16(g, x, (off[-1], C), off, pls, H, dy_dx=dd)
    dd4 = dd.reshape(B, L, D, C)
    gi = np.zeros((B, D), np.float16)
    for b in range(B):
        for d in range(D):
            acc = np.float16(0)
            for l in range(L):
                for ch in range(C):
                    acc = np.float16(np.float32(acc) + np.float32(np.float16(np.float32(g[b, l * C + ch]) * np.float32(dd4[b, l, d, ch]))))
            gi[b, d] = acc
    assert np.array_equal(r2["grad_inputs"].view(np.uint16), gi.view(np.uint16))
    assert np.array_equal(r2["grad_embeddings"], r["grad_embeddings"])


def _composite_ref64(sig, rgb, dl, rays, unc=None, T_thresh=1e-4):
    N = rays.shape[0]
    ws, dep, img, us = np.zeros(N), np.zeros(N), np.zeros((N, 3)), np.zeros(N)
    for n in range(N):
        i, o, c = rays[n]
        T = 1.0
        for s in range(o, o + c):
            a = 1 - np.exp(-float(sig[s]) * float(dl[s, 0]))
            w = a * T
            ws[i] += w
            dep[i] += w * dl[s, 1]
            img[i] += w * rgb[s]
            if unc is not None:
                us[i] += w * unc[s]
            T *= 1 - a
            if T < T_thresh:
                break
    return ws, dep, img, us


def _random_rays(rng, N, max_c):
    counts = rng.integers(0, max_c, N)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    ids = rng.permutation(N)
    return np.stack([ids, offs, counts], 1).astype(np.int32), int(counts.sum())


def test_composite_train_forward_and_backward():
    rng = np.random.default_rng(9)
    rays, M = _random_rays(rng, 40, 30)
    sig = rng.uniform(0, 60, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    dl = np.stack([rng.uniform(0.005, 0.03, M), np.cumsum(rng.uniform(0.01, 0.03, M))], 1).astype(np.float32)
    a0, a1, unc = [rng.uniform(0, 1, M).astype(np.float32) for _ in range(3)]
    fwd = O.composite_rays_train_forward("triplane", sig, rgb, dl, rays, a0, a1, unc)
    ws, dep, img, us = _composite_ref64(sig, rgb, dl, rays, unc)
    assert np.allclose(fwd["weights_sum"], ws, atol=1e-5) and np.allclose(fwd["depth"], dep, atol=1e-4)
    assert np.allclose(fwd["image"], img, atol=1e-5) and np.allclose(fwd["unc_sum"], us, atol=1e-5)
    # ambient channels are UNWEIGHTED sums over the visited samples (raymarching.cu:1942-1943)
    for n in range(rays.shape[0]):
        i, o, c = rays[n]
        T, vis = 1.0, 0
        for s in range(o, o + c):
            vis += 1
            T *= np.exp(-float(sig[s]) * float(dl[s, 0]))
            if T < 1e-4:
                break
        assert fwd["amb0_sum"][i] == pytest.approx(a0[o:o + vis].astype(np.float64).sum(), abs=1e-4)
    # analytic backward vs float64 central differences of the float64 forward
    gws, gimg, gu = rng.normal(size=40).astype(np.float32), rng.normal(size=(40, 3)).astype(np.float32), rng.normal(size=40).astype(np.float32)
    g = O.composite_rays_train_backward("triplane", dict(grad_weights_sum=gws, grad_image=gimg, grad_amb0_sum=np.ones(40, np.float32),
                                                        grad_amb1_sum=2 * np.ones(40, np.float32), grad_unc_sum=gu),
                                        sig, rgb, dl, rays, fwd, a0, a1, unc)

    def loss(s_):
        w_, _, i_, u_ = _composite_ref64(s_, rgb, dl, rays, unc, T_thresh=-1.0)
        return np.sum(w_ * gws) + np.sum(i_ * gimg) + np.sum(u_ * gu)

    # compare only on rays that never hit the early-termination threshold (the analytic form assumes the full sum)
    sig_small = (sig * 0.05).astype(np.float32)
    fwd2 = O.composite_rays_train_forward("triplane", sig_small, rgb, dl, rays, a0, a1, unc)
    g2 = O.composite_rays_train_backward("triplane", dict(grad_weights_sum=gws, grad_image=gimg, grad_amb0_sum=np.ones(40, np.float32),
                                                         grad_amb1_sum=2 * np.ones(40, np.float32), grad_unc_sum=gu),
                                         sig_small, rgb, dl, rays, fwd2, a0, a1, unc)
    base = sig_small.astype(np.float64)

    def loss2(s_):
        w_, _, i_, u_ = _composite_ref64(s_, rgb, dl, rays, unc, T_thresh=-1.0)
        return np.sum(w_ * gws) + np.sum(i_ * gimg) + np.sum(u_ * gu)

    for s in rng.choice(M, 25, replace=False):
        hp, hm = base.copy(), base.copy()
        hp[s] += 1e-4
        hm[s] -= 1e-4
        fd = (loss2(hp) - loss2(hm)) / 2e-4
        assert g2["grad_sigmas"][s] == pytest.approx(fd, abs=2e-3 + 2e-3 * abs(fd))
    assert np.allclose(g2["grad_rgbs"].sum(), g2["grad_rgbs"].sum())  # finite
    assert np.all(g["grad_amb0"][g["grad_rgbs"][:, 0] != 0] == 1.0) and np.all(np.isin(g["grad_amb1"], [0.0, 2.0]))


def test_composite_inference_equals_train_composite_when_chunked():
    """resuming with T = 1 - weights_sum over several n_step chunks must reproduce one straight pass"""
    rng = np.random.default_rng(10)
    N, S = 16, 12
    sig = rng.uniform(0, 30, (N, S)).astype(np.float32)
    rgb = rng.uniform(0, 1, (N, S, 3)).astype(np.float32)
    dt = np.full((N, S), 0.02, np.float32)
    tt = np.cumsum(dt, 1).astype(np.float32) + 2
    acc = dict(weights_sum=np.zeros(N, np.float32), depth=np.zeros(N, np.float32), image=np.zeros((N, 3), np.float32))
    alive = np.arange(N, dtype=np.int32)
    rays_t = np.full(N, 2.0, np.float32)
    for c0 in range(0, S, 4):
        ids = alive.copy()
        n = len(ids)
        dl = np.stack([dt[ids, c0:c0 + 4], tt[ids, c0:c0 + 4]], -1).reshape(-1, 2)
        O.composite_rays("plain", n, 4, ids, rays_t, sig[alive, c0:c0 + 4].reshape(-1), rgb[alive, c0:c0 + 4].reshape(-1, 3), dl,
                         acc["weights_sum"], acc["depth"], acc["image"], T_thresh=1e-4)
        alive = ids[ids >= 0]
        if len(alive) == 0:
            break
    for n in range(N):
        w = 0.0
        img = np.zeros(3)
        for s in range(S):
            a = 1 - np.exp(-float(sig[n, s]) * 0.02)
            T = 1 - w
            w += a * T
            img += a * T * rgb[n, s]
            if T < 1e-4:
                break
        assert acc["weights_sum"][n] == pytest.approx(w, abs=1e-5)
        assert np.allclose(acc["image"][n], img, atol=1e-5)


def test_march_properties():
    from conftest import ellipsoid_bitfield, synthetic_camera
    from oracle.head import get_rays
    H = W = 32
    pose, intr = synthetic_camera(H, W)
    ro, rd = get_rays(pose, intr, H, W)
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.05)
    ones = np.full(128 ** 3 // 8, 255, np.uint8)
    N = ro.shape[0]
    # all-ones grid: every step is a sample, dt == dt_min == 2 sqrt3 / max_steps
    x, d, dl, rays = O.march_rays_train(ro, rd, 1.0, ones, 1, 128, nears, fars, None, -1, None, 128, True, 1 / 256, 64)
    cnt = rays[:, 2]
    s3 = np.float32(1.7320508075688772)
    dt_min = min(np.float32(2 * s3 * 1 / 128), np.float32(2 * s3 / 64))  # min(dt_max, 2 sqrt3 / max_steps), raymarching.cu:386-387
    valid = dl[:, 0] != 0
    assert np.all(dl[valid, 0] == dt_min)
    hit = nears < 1e30
    exp_cnt = np.where(hit, np.minimum(np.ceil((fars - nears) / dt_min - 1e-4), 64), 0)
    assert np.all(np.abs(cnt - exp_cnt) <= 1)
    assert x.shape[0] % 128 == 0 and x.shape[0] - cnt.sum() <= 128 and x.shape[0] > cnt.sum() - 1
    assert np.all(np.abs(x[valid]) <= 1.0)
    # samples lie on their rays: x = o + t d with t = t_after - dt
    for n in rng_subset(N):
        i, o, c = rays[n]
        if c:
            t = dl[o:o + c, 1] - dl[o:o + c, 0]
            assert np.allclose(x[o:o + c], ro[i] + t[:, None] * rd[i], atol=2e-6)
            assert np.all(np.diff(dl[o:o + c, 1]) > 0)
    # empty grid -> nothing; ellipsoid -> only rays through it, far fewer samples
    zeros = np.zeros_like(ones)
    _, _, _, r0 = O.march_rays_train(ro, rd, 1.0, zeros, 1, 128, nears, fars, None, -1, None, 128, True, 1 / 256, 64)
    assert r0[:, 2].sum() == 0
    ell, grid = ellipsoid_bitfield()
    assert 0.02 < grid.mean() < 0.04
    xe, _, dle, re = O.march_rays_train(ro, rd, 1.0, ell, 1, 128, nears, fars, None, -1, None, 128, True, 1 / 256, 64)
    ve = dle[:, 0] != 0
    assert 0 < re[:, 2].sum() < cnt.sum() * 0.6
    q = (xe[ve] / np.array([0.35, 0.45, 0.35], np.float32)) ** 2
    assert np.all(q.sum(1) < 1.35)  # samples sit in occupied cells (cell size 1/64 slack)
    # inference march in chunks reproduces the training march sample sequence
    alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    xs, _, ds = O.march_rays(N, 4, alive, rays_t, ro, rd, 1.0, ell, 1, 128, nears, fars, 128, None, 1 / 256, 64)
    by_ray = {int(r[0]): r for r in re}
    for n in rng_subset(N):
        c = min(4, int(by_ray[n][2]))
        o = int(by_ray[n][1])
        assert np.array_equal(xs[n * 4:n * 4 + c], xe[o:o + c]) and np.array_equal(ds[n * 4:n * 4 + c], dle[o:o + c])
        if c < 4:
            assert np.all(ds[n * 4 + c:n * 4 + 4] == 0)


def rng_subset(N, k=64):
    return np.random.default_rng(11).choice(N, min(k, N), replace=False)


def test_march_train_overflow_drops_rays_like_reference():
    from conftest import synthetic_camera
    from oracle.head import get_rays
    pose, intr = synthetic_camera(16, 16)
    ro, rd = get_rays(pose, intr, 16, 16)
    nears, fars = O.near_far_from_aabb(ro, rd, np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32), 0.05)
    ones = np.full(128 ** 3 // 8, 255, np.uint8)
    ctr = np.zeros(2, np.int32)
    x, d, dl, rays = O.march_rays_train(ro, rd, 1.0, ones, 1, 128, nears, fars, ctr, 1000, None, 128, False, 1 / 256, 32)
    assert x.shape[0] == 1024  # mean_count padded up to a multiple of 128 (raymarching.py:226-228)
    assert ctr[1] == 256 and ctr[0] == rays[:, 2].sum() > 1024
    dropped = rays[:, 1] + rays[:, 2] > 1024
    assert dropped.any() and not dropped.all()
    comp = O.composite_rays_train_forward("ambient", np.ones(1024, np.float32), np.ones((1024, 3), np.float32), dl, rays,
                                          amb0=np.ones(1024, np.float32))
    assert np.all(comp["weights_sum"][rays[dropped, 0]] == 0) and np.all(comp["weights_sum"][rays[~dropped & (rays[:, 2] > 0), 0]] > 0)


def test_occupancy_update_restatement_small():
    """oracle/occupancy.py on a 4^3 grid with a constant-density stand-in: EMA / untrained / dilation / threshold logic"""
    from oracle import occupancy as OC
    G = 4
    grid = np.zeros((1, G ** 3), np.float32)
    grid[0, 5] = -1.0
    grid[0, 7] = 3.0
    calls = {}
    orig_density, orig_encode = OC.density, OC.encode_x
    try:
        OC.encode_x = lambda spec, x, P: x
        def fake_density(spec, P, enc_x, enc_a, eye):
            calls["xyz"] = enc_x
            return dict(sigma=(enc_x[:, 0] > 0).astype(np.float32))   # density 1 on the +x half
        OC.density = fake_density
        noise = np.full((1, G ** 3, 3), 0.5, np.float32)               # zero offset
        mean, thresh, bits = OC.update_density_grid(None, None, grid, None, None, 1.0, noise, decay=0.5, density_thresh=0.25)
    finally:
        OC.density, OC.encode_x = orig_density, orig_encode
    xyz = calls["xyz"]
    assert xyz.shape == (64, 3) and np.allclose(xyz.min(), -0.75) and np.allclose(xyz.max(), 0.75)   # (2c/3 - 1) * (1 - 1/4)
    coords = np.stack(np.meshgrid(*[np.arange(G)] * 3, indexing="ij"), -1).reshape(-1, 3)
    m = O.morton3D(coords)
    expect = np.where(coords[:, 0] >= 1, 1.0, 0.0)                     # x >= 2 has density 1; dilation reaches x = 1
    for c, mi in zip(coords, m):
        if mi == 5:
            assert grid[0, mi] == -1.0                                  # untrained stays
        elif mi == 7:
            assert grid[0, mi] == max(3.0 * 0.5, expect[(coords == c).all(1)][0])
        else:
            assert grid[0, mi] == expect[(coords == c).all(1)][0]
    assert thresh == 0.25 and mean == pytest.approx(np.clip(grid, 0, None).mean())
    assert np.array_equal(np.unpackbits(bits, bitorder="little")[:64].astype(bool), grid[0] > 0.25)


def test_torso_grid_sample_restatement_matches_torch():
    """oracle/torso.py:grid_sample_2d (run_torso's occupancy lookup, renderer.py:604-605) vs torch's own F.grid_sample on the CPU"""
    import torch
    from oracle.torso import grid_sample_2d
    rng = np.random.default_rng(0)
    g = rng.uniform(0, 1, (128, 128)).astype(np.float32)
    c = rng.uniform(-1.05, 1.05, (5000, 2)).astype(np.float32)
    c[:3] = [[-1, -1], [1, 1], [0, 0]]
    ref = torch.nn.functional.grid_sample(torch.from_numpy(g).view(1, 1, 128, 128), torch.from_numpy(c).view(1, -1, 1, 2),
                                          mode="bilinear", padding_mode="zeros", align_corners=True).view(-1).numpy()
    assert np.abs(grid_sample_2d(g, c) - ref).max() <= 1e-6
