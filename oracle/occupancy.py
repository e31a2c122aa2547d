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
