"""Occupancy-grid maintenance on device: the head branch of `NeRFRenderer.update_extra_state`
(/root/reference/nerf_triplane/renderer.py:699-766) without its Python cascade loop, boolean-mask EMA and `.item()` syncs.

    mean, thresh = update_density_grid(head, density_grid, density_bitfield, enc_a, eye, bound=1.0)

`density_grid` [cascade, G^3] (Morton order, -1 = untrained) and `density_bitfield` [cascade * G^3 / 8] are updated in place,
exactly as the reference's attributes of the same names; `mean` / `thresh` are 0-d device tensors (the reference's
`self.mean_density` and the threshold it packs with), ready in stream order.
"""
import math

import torch

from ._util import call, ptr, stream


@torch.no_grad()
def update_density_grid(head, density_grid, density_bitfield, enc_a, eye=None, bound=1.0, decay=0.95, density_thresh=0.01,
                        density_scale=1.0, noise=None):
    cascade, cells = density_grid.shape
    G = round(cells ** (1 / 3))
    if G ** 3 != cells or density_grid.dtype != torch.float32 or not density_grid.is_contiguous():
        raise RuntimeError("density_grid must be a contiguous float32 [cascade, grid_size^3] tensor")
    if cascade != 1 + math.ceil(math.log2(bound)):
        raise RuntimeError("cascade does not match bound (renderer.py:93)")
    dev = density_grid.device
    n = cascade * cells
    if noise is None:   # one torch.rand per cascade, in the order the reference draws them (S = grid_size: one block per cascade)
        noise = torch.stack([torch.rand(cells, 3, dtype=torch.float32, device=dev) for _ in range(cascade)])
    noise = noise.to(dev, torch.float32).contiguous()
    if noise.numel() != n * 3:
        raise RuntimeError("noise must hold cascade * grid_size^3 * 3 values")
    xyzs = torch.empty(n, 3, dtype=torch.float32, device=dev)
    call("lz_density_grid_points", ptr(noise), cascade, G, float(bound), ptr(xyzs), stream())
    dirs = torch.zeros(n, 3, dtype=torch.float32, device=dev)
    dirs[:, 2] = 1.0   # the density does not depend on the view direction; the fused head still wants one
    sigma = head.forward(xyzs, dirs, enc_a, None, eye, testing=True)[0]
    stats = torch.empty(2, dtype=torch.float32, device=dev)
    workspace = torch.empty((n + 255) // 256, dtype=torch.float32, device=dev)
    call("lz_density_grid_update", ptr(sigma), float(density_scale), float(decay), float(density_thresh), cascade, G, ptr(density_grid),
         ptr(density_bitfield), ptr(stats), ptr(workspace), stream())
    # both tensors were written through raw pointers: tell torch (version counters), so that whoever keys derived data on them --
    # TriplaneRenderer.occupied_bounds() on the bitfield -- sees the change
    torch.autograd.graph.increment_version(density_grid)
    torch.autograd.graph.increment_version(density_bitfield)
    return stats[0], stats[1]


@torch.no_grad()
def mark_untrained_grid(density_grid, poses, intrinsic, bound=1.0, return_count=False):
    """`NeRFRenderer.mark_untrained_grid` (renderer.py:633-695): cells of `density_grid` [cascade, G^3] that none of the training cameras
    `poses` [B,4,4] (c2w) with `intrinsic` (fx, fy, cx, cy) sees are set to -1 in place (one launch; the reference runs a 5-level Python
    loop of batched matmuls).  Such cells are skipped by `update_density_grid` and never marched."""
    cascade, cells = density_grid.shape
    G = round(cells ** (1 / 3))
    if G ** 3 != cells or density_grid.dtype != torch.float32 or not density_grid.is_contiguous():
        raise RuntimeError("density_grid must be a contiguous float32 [cascade, grid_size^3] tensor")
    poses = torch.as_tensor(poses).to(density_grid.device, torch.float32).reshape(-1, 4, 4).contiguous()
    fx, fy, cx, cy = [float(v) for v in intrinsic]
    count = torch.empty(cascade, cells, dtype=torch.int32, device=density_grid.device) if return_count else None
    call("lz_mark_untrained_grid", ptr(poses), poses.shape[0], fx, fy, cx, cy, cascade, G, float(bound), ptr(density_grid), ptr(count), stream())
    return count


@torch.no_grad()
def update_density_grid_torso(torso, density_grid_torso, poses, ind_code=None, decay=0.95, density_thresh=0.01, noise=None, enc_anchor=None):
    """Torso half of `update_extra_state` (renderer.py:772-808): alpha of `forward_torso` at one jittered point per cell of the
    G x G torso grid, 5 x 5 max pool, EMA in place.  `torso`: lzzx_nerf_amd.torso.FusedTorso.  Returns device scalars
    (mean_density_torso, min(mean, density_thresh)); the second is the threshold `run_torso` masks with (renderer.py:603)."""
    cells = density_grid_torso.numel()
    G = round(cells ** 0.5)
    if G * G != cells or density_grid_torso.dtype != torch.float32 or not density_grid_torso.is_contiguous():
        raise RuntimeError("density_grid_torso must be a contiguous float32 tensor of grid_size^2 values")
    dev = density_grid_torso.device
    if noise is None:
        noise = torch.rand(cells, 2, dtype=torch.float32, device=dev)
    noise = noise.to(dev, torch.float32).contiguous()
    xys = torch.empty(cells, 2, dtype=torch.float32, device=dev)
    call("lz_density_grid_torso_points", ptr(noise), G, ptr(xys), stream())
    alpha = torso(xys, poses, ind_code, enc_anchor=enc_anchor)[0]
    stats = torch.empty(2, dtype=torch.float32, device=dev)
    workspace = torch.empty((cells + 255) // 256, dtype=torch.float32, device=dev)
    call("lz_density_grid_torso_update", ptr(alpha), float(decay), float(density_thresh), G, ptr(density_grid_torso), ptr(stats),
         ptr(workspace), stream())
    return stats[0], stats[1]
