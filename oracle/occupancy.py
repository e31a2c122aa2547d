"""TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  CPU restatement of the head branch of NeRFRenderer.update_extra_state
(/root/reference/nerf_triplane/renderer.py:699-766), numpy float32 in the reference's operation order."""
import numpy as np

from . import oracle as O
from .head import density, encode_x

F32 = np.float32


def update_density_grid(spec, P, density_grid, enc_a, eye, bound, noise, decay=0.95, density_thresh=0.01, density_scale=1.0):
    """density_grid [C, G^3] f32 (Morton order) is updated in place.  noise [C, G^3, 3] plays torch.rand_like (:751).
    Returns (mean_density, threshold used, bitfield)."""
    C, cells = density_grid.shape
    G = round(cells ** (1 / 3))
    ax = np.arange(G, dtype=np.int32)
    xx, yy, zz = np.meshgrid(ax, ax, ax, indexing="ij")                        # custom_meshgrid, :739
    coords = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], 1)
    indices = O.morton3D(coords).astype(np.int64)                              # :742
    xyzs = F32(2) * coords.astype(F32) / F32(G - 1) - F32(1)                   # :743
    tmp = np.zeros_like(density_grid)
    for cas in range(C):
        bc = min(2 ** cas, bound)
        half = bc / G
        cas_xyzs = xyzs * F32(bc - half)                                       # :749
        cas_xyzs = cas_xyzs + (noise[cas].astype(F32) * F32(2) - F32(1)) * F32(half)   # :751
        sig = density(spec, P, encode_x(spec, cas_xyzs, P), enc_a, eye)["sigma"].reshape(-1).astype(F32)   # :753
        tmp[cas, indices] = sig * F32(density_scale)                           # :754-757
    tmp = O.morton3D_dilation(tmp)                                             # :760
    valid = (density_grid >= 0) & (tmp >= 0)                                   # :763
    density_grid[valid] = np.maximum(density_grid[valid] * F32(decay), tmp[valid])
    mean = float(np.mean(np.clip(density_grid, 0, None), dtype=np.float64))    # :765 (torch sums in f32; the order is its own)
    thresh = min(mean, density_thresh)                                         # :770
    return mean, thresh, O.packbits(density_grid, thresh)


def update_density_grid_torso(P, density_grid_torso, enc_anchor, ind_code, noise, decay=0.95, torso_shrink=0.8):
    """torso half of update_extra_state (renderer.py:772-808); density_grid_torso [G*G] is updated in place; returns its mean"""
    from .torso import forward_torso
    cells = density_grid_torso.size
    G = round(cells ** 0.5)
    ax = np.arange(G, dtype=np.int32)
    xx, yy = np.meshgrid(ax, ax, indexing="ij")                                           # custom_meshgrid(xs, ys), :790
    coords = np.stack([xx.ravel(), yy.ravel()], 1)
    indices = (coords[:, 1].astype(np.int64) * G + coords[:, 0])                         # xy transposed, :792
    xys = F32(2) * coords.astype(F32) / F32(G - 1) - F32(1)                               # :793
    xys = xys * F32(1 - 1 / G)                                                            # :794
    xys = xys + (noise.astype(F32) * F32(2) - F32(1)) * F32(1 / G)                        # :796
    alphas, _, _ = forward_torso(P, xys, enc_anchor, ind_code, torso_shrink)             # :798
    tmp = np.zeros(cells, F32)
    tmp[indices] = alphas[:, 0]
    t2 = tmp.reshape(G, G)
    pad = np.full((G + 4, G + 4), -np.inf, F32)
    pad[2:-2, 2:-2] = t2
    dil = np.max(np.stack([pad[dy:dy + G, dx:dx + G] for dy in range(5) for dx in range(5)]), 0)   # F.max_pool2d(k=5, s=1, p=2), :804
    density_grid_torso[:] = np.maximum(density_grid_torso * F32(decay), dil.reshape(-1))              # :806
    return float(np.mean(density_grid_torso, dtype=np.float64))                                     # :807


def mark_untrained_grid(density_grid, poses, intrinsic, bound, return_margin=False):
    """NeRFRenderer.mark_untrained_grid (renderer.py:633-695): cells no camera sees get density -1, in place.
    density_grid [C, G^3] f32 Morton-ordered; poses [B,4,4] c2w; intrinsic (fx, fy, cx, cy).  The block / batch splitting (S = 64) of the
    reference only bounds memory; counts are per cell.  return_margin: also the smallest |lhs - rhs| of the three frustum tests per cell
    and cascade (cells decided by less than rounding can legitimately differ between matmul orders)."""
    C, cells = density_grid.shape
    G = round(cells ** (1 / 3))
    fx, fy, cx, cy = [float(v) for v in intrinsic]
    poses = np.asarray(poses, F32).reshape(-1, 4, 4)
    ax = np.arange(G, dtype=np.int32)
    xx, yy, zz = np.meshgrid(ax, ax, ax, indexing="ij")                                   # custom_meshgrid, :662
    coords = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], 1)
    indices = O.morton3D(coords).astype(np.int64)                                         # :665
    world = F32(2) * coords.astype(F32) / F32(G - 1) - F32(1)                             # :666
    count = np.zeros((C, cells), np.int64)
    margin = np.full((C, cells), np.inf)
    for cas in range(C):
        bc = min(2 ** cas, bound)
        half = bc / G
        cw = world * F32(bc - half)                                                       # :672
        for pose in poses:
            v = cw - pose[:3, 3][None, :]                                                 # :679
            cam = (v[:, 0:1] * pose[0:1, :3] + v[:, 1:2] * pose[1:2, :3]) + v[:, 2:3] * pose[2:3, :3]   # `@ poses[:, :3, :3]`, :680
            cam = cam.astype(F32)
            rx = (F32(cx / fx) * cam[:, 2] + F32(half * 2)).astype(F32)                   # :684 (python float scalars: cx / fx in double, then f32)
            ry = (F32(cy / fy) * cam[:, 2] + F32(half * 2)).astype(F32)
            m = (cam[:, 2] > 0) & (np.abs(cam[:, 0]) < rx) & (np.abs(cam[:, 1]) < ry)     # :683-686
            count[cas, indices] += m
            mg = np.minimum(np.abs(cam[:, 2]), np.minimum(np.abs(np.abs(cam[:, 0]) - rx), np.abs(np.abs(cam[:, 1]) - ry)))
            margin[cas, indices] = np.minimum(margin[cas, indices], mg)
    density_grid[count == 0] = -1                                                         # :692
    return (count, margin) if return_margin else count
