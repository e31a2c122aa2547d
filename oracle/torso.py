"""TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  CPU restatement of the torso branch: run_torso's masked query
(/root/reference/nerf_triplane/renderer.py:572-631) and forward_torso (/root/reference/nerf_triplane/network.py:170-205)."""
import numpy as np

from . import oracle as O

F32 = np.float32


def grid_sample_2d(grid, coords):
    """F.grid_sample(grid[1,1,G,G], coords[1,N,1,2], mode='bilinear', padding_mode='zeros', align_corners=True) -> [N]
    in torch's CUDA operation order (GridSampler.cuh): unnormalise ((c + 1) / 2) * (size - 1); weights from the corner
    coordinates; `out += value * weight` in nw, ne, sw, se order (fused by nvcc's default -fmad)."""
    G = grid.shape[0]
    ix = ((coords[:, 0].astype(F32) + F32(1)) / F32(2)) * F32(G - 1)
    iy = ((coords[:, 1].astype(F32) + F32(1)) / F32(2)) * F32(G - 1)
    x0f, y0f = np.floor(ix), np.floor(iy)
    x1f, y1f = x0f + F32(1), y0f + F32(1)
    w = [(x1f - ix) * (y1f - iy), (ix - x0f) * (y1f - iy), (x1f - ix) * (iy - y0f), (ix - x0f) * (iy - y0f)]
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
    out = np.zeros(coords.shape[0], F32)
    for (dx, dy), wi in zip([(0, 0), (1, 0), (0, 1), (1, 1)], w):
        xx, yy = x0 + dx, y0 + dy
        ok = (xx >= 0) & (yy >= 0) & (xx < G) & (yy < G)
        v = np.where(ok, grid[np.clip(yy, 0, G - 1), np.clip(xx, 0, G - 1)], F32(0)).astype(F32)
        out = O.fma(v, wi.astype(F32), out)
    return out


def forward_torso(P, x, enc_anchor, ind_code, torso_shrink=0.8):
    """x [M,2] in [-1,1] (already masked); enc_anchor [42]; ind_code [ind] or None.  Returns alpha [M,1], color [M,3], dx [M,2].
    Chain order of the two first layers: frame-constant inputs first (csrc/lz_torso.hip), then per-pixel inputs, natural order."""
    x = x.astype(F32) * F32(torso_shrink)                                               # network.py:176
    enc_x = O.freq_encode_forward(x, 8)                                                 # [M, 34]
    M = x.shape[0]
    const = [np.asarray(enc_anchor, F32).reshape(1, -1)]
    if ind_code is not None:
        const.append(np.asarray(ind_code, F32).reshape(1, -1))
    const = np.repeat(np.concatenate(const, 1), M, 0)
    h = np.ascontiguousarray(np.concatenate([enc_x, const], 1))                         # network.py:186-189
    K0 = h.shape[1]
    order0 = list(range(34, K0)) + list(range(34))
    d = O.linear(h, P["torso_deform_net.net.0.weight"], order0, relu=True)
    d = O.linear(d, P["torso_deform_net.net.1.weight"], relu=True)
    dx = O.linear(d, P["torso_deform_net.net.2.weight"])                                # [M, 2]
    xx = np.clip(x + dx, F32(-1), F32(1))                                               # network.py:193
    x01 = (xx + F32(1)) / F32(2)                                                        # grid.py:143, bound = 1
    pls = np.exp2(np.log2(2048 / 16) / 15)
    gx, _ = O.grid_encode_forward(x01, P["torso_encoder.embeddings"], P["torso_encoder.offsets"], pls, 16, False, 1)
    h2 = np.ascontiguousarray(np.concatenate([gx, h], 1))                               # network.py:198
    K1 = h2.shape[1]
    order1 = list(range(66, K1)) + list(range(66))
    t = O.linear(h2, P["torso_net.net.0.weight"], order1, relu=True)
    t = O.linear(t, P["torso_net.net.1.weight"], relu=True)
    t = O.linear(t, P["torso_net.net.2.weight"])
    out = O.unary("sigmoid", t) * F32(1 + 2 * 0.001) - F32(0.001)                       # network.py:202-203
    return out[:, :1], out[:, 1:], dx


def run_torso(P, bg_coords, enc_anchor, ind_code, density_grid=None, density_thresh=0.0, torso_shrink=0.8):
    """renderer.py:603-617: occupancy mask, masked query, scatter into zero tensors"""
    N = bg_coords.shape[0]
    alpha, color, deform = np.zeros((N, 1), F32), np.zeros((N, 3), F32), np.zeros((N, 2), F32)
    mask = np.ones(N, bool)
    if density_grid is not None:
        G = round(density_grid.size ** 0.5)
        mask = grid_sample_2d(density_grid.reshape(G, G).astype(F32), bg_coords) > F32(density_thresh)
    if mask.any():
        a, c, dx = forward_torso(P, bg_coords[mask], enc_anchor, ind_code, torso_shrink)
        alpha[mask], color[mask], deform[mask] = a, c, dx
    return alpha, color, deform, mask


def encode_anchor(P, pose):
    """network.py:179-183: the three anchor points through inverse(pose^T), perspective divide, frequency encoding (deg 3) -> [1, 42]"""
    A = np.asarray(P["anchor_points"], F32)                                               # [3, 4]
    inv = np.linalg.inv(np.asarray(pose, F32).reshape(4, 4).T.astype(np.float64))          # torch inverts in f32 (LU); pinned to 1e-5 by the fixture
    w = (A.astype(np.float64) @ inv).astype(F32)
    w = (w[:, :2] / w[:, 3:4] / w[:, 2:3]).astype(F32).reshape(1, -1)
    return O.freq_encode_forward(np.ascontiguousarray(w), 3)
