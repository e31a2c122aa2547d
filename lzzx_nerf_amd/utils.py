"""Ray generation with the reference's signature and result dict: `nerf_triplane.utils.get_rays` / `get_bg_coords`
(/root/reference/nerf_triplane/utils.py:217-312) on the gfx950 kernels (csrc/lz_raymarch.hip: lz_k_get_rays, lz_k_bg_coords).

    r = get_rays(poses [B,4,4], intrinsics [fx,fy,cx,cy], H, W, N=-1, patch_size=1, rect=None)
    r['rays_o'], r['rays_d'] [B,N,3] f32;  r['inds'] [B,N] int64;  r['i'], r['j'] [B,N] pixel-centre coordinates

Pixel selection happens on the device and never synchronises: the random branches draw from torch's generator with the same calls, in
the same order, as the reference (so a seeded run selects the same pixels), the rect branch computes its index list arithmetically
instead of `torch.where(mask)` (which copies a count to the host).  The kernel then reads `inds`; i, j, directions and origins are
never materialised for pixels that were not selected.
"""
import torch

from ._util import call, ptr, stream


def select_pixels(H, W, N=-1, patch_size=1, rect=None, device="cuda"):
    """-> int64 [N] flat pixel indices (row * W + col) for the N > 0 branches of get_rays (utils.py:249-285), or None for the
    full image.  Pure torch (device-agnostic), split out so that the index rules can be checked without a GPU."""
    if rect is not None:                       # utils.py:239-241 overrides N; rows xmin..xmax of columns ymin..ymax (:278)
        xmin, xmax, ymin, ymax = [int(v) for v in rect]
        N = (xmax - xmin) * (ymax - ymin)
    if N <= 0:
        return None
    N = min(N, H * W)
    if patch_size > 1:                         # :252-271: random top-left corners, patch_size x patch_size pixels each
        num_patch = N // (patch_size ** 2)
        x0 = torch.randint(0, H - patch_size, size=[num_patch], device=device)
        y0 = torch.randint(0, W - patch_size, size=[num_patch], device=device)
        d = torch.arange(patch_size, device=device)
        rows = x0[:, None, None] + d[None, :, None]          # offset order: row offset outer, column offset inner
        cols = y0[:, None, None] + d[None, None, :]
        return (rows * W + cols).reshape(-1)
    if rect is not None:                       # :274-281: ascending flat indices of the rectangle, clipped to the image like a slice
        r = torch.arange(max(xmin, 0), min(xmax, H), device=device)
        c = torch.arange(max(ymin, 0), min(ymax, W), device=device)
        return (r[:, None] * W + c[None, :]).reshape(-1)
    return torch.randint(0, H * W, size=[N], device=device)  # :284, may repeat


def get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, rect=None):
    poses = torch.as_tensor(poses)
    if poses.requires_grad:
        raise RuntimeError("get_rays: no gradient path to the poses (opt.train_camera); detach them")
    poses = poses.reshape(-1, 4, 4).float().contiguous()
    if not poses.is_cuda:
        raise RuntimeError("poses must be a CUDA tensor")
    dev, B = poses.device, poses.shape[0]
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    sel = select_pixels(H, W, N, patch_size, rect, dev)
    n = H * W if sel is None else sel.shape[0]
    kw = dict(dtype=torch.float32, device=dev)
    rays_o, rays_d = torch.empty(B, n, 3, **kw), torch.empty(B, n, 3, **kw)
    i, j = torch.empty(B, n, **kw), torch.empty(B, n, **kw)
    call("lz_get_rays", ptr(poses), fx, fy, cx, cy, int(H), int(W), B, n, ptr(sel), ptr(rays_o), ptr(rays_d), ptr(i), ptr(j), stream())
    rect_branch = sel is not None and rect is not None and patch_size <= 1
    if sel is None:
        sel = torch.arange(H * W, device=dev)
    inds = sel.unsqueeze(0) if rect_branch else sel.expand(B, n)   # the rect branch returns [1, N] (:281), the others [B, N]
    return {"i": i, "j": j, "inds": inds, "rays_o": rays_o, "rays_d": rays_d}


def frame_rays(pose, intrinsics, H, W, pixels=None):
    """rays of one frame: pose [4,4] -> rays_o, rays_d [n, 3] for every pixel in order (pixels None) or for the int64 device index list
    `pixels` (a rank's tile of a ray-sharded frame, lzzx_nerf_amd.dist.tile_pixels); no i / j / inds outputs.  What the renderer and
    bench.py consume."""
    pose = pose.reshape(1, 4, 4).float().contiguous()
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    n = H * W if pixels is None else pixels.shape[0]
    rays_o = torch.empty(n, 3, dtype=torch.float32, device=pose.device)
    rays_d = torch.empty(n, 3, dtype=torch.float32, device=pose.device)
    call("lz_get_rays", ptr(pose), fx, fy, cx, cy, int(H), int(W), 1, n, ptr(pixels), ptr(rays_o), ptr(rays_d), None, None, stream())
    return rays_o, rays_d


def get_bg_coords(H, W, device):
    """[1, H*W, 2] in [-1, 1] (utils.py:217-223)"""
    out = torch.empty(1, H * W, 2, dtype=torch.float32, device=device)
    call("lz_bg_coords", int(H), int(W), ptr(out), stream())
    return out
