"""Seeded synthetic workload of SURVEY 8(d): camera, scene, parameters.  Shared by bench.py, __graft_entry__.smoke() and the
tests; nothing here touches the CPU checker (the occupancy bitfield is built with the HIP morton3D / packbits operators).

Camera: c2w = identity rotation, t = (0, 0, -3.35) (the reference GUI radius, train.py:96), fovy 21.24 deg (train.py:97),
intrinsics convention of provider.py:614-631.  Scene: bound 1, 128^3 grid, head-like ellipsoid (semi-axes .35/.45/.35) or all ones.
Parameters: MLP weights = the reference's own torch init under seed 0 (the committed fixture state-dict), tables U(-1, 1).
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_python.npz")


def load_golden():
    return np.load(GOLDEN, allow_pickle=False)


def synthetic_camera(H, W):
    """identity rotation, t = (0, 0, -3.35), fovy 21.24 deg -> (pose [4,4] f32, [fl_x, fl_y, cx, cy])"""
    fl = H / (2 * np.tan(np.radians(21.24) / 2))
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = -3.35
    return pose, [fl, fl, W / 2, H / 2]


def orbit_pose(k):
    """head pose of frame k of a clip: camera on a circle of radius 3.35 about the vertical axis, 0.01 rad per frame, frame 0 frontal"""
    th = 0.01 * k
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], dtype=np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R
    pose[:3, 3] = R @ np.array([0, 0, -3.35], dtype=np.float32)
    return pose


def make_params(golden=None, seed_tables=1234):
    """state-dict-shaped numpy weights: MLP weights from the reference fixture, tables regenerated from the seed used by
    tests/golden/make_golden.py"""
    if golden is None:
        golden = load_golden()
    P = {k[3:]: golden[k] for k in golden.files if k.startswith("sd/")}
    rng = np.random.default_rng(seed_tables)
    n = int(P["encoder_xy.offsets"][-1])
    for name in ("xy", "yz", "xz"):
        P[f"encoder_{name}.embeddings"] = rng.uniform(-1, 1, (n, 1)).astype(np.float32)
    return P


def ellipsoid_grid(grid_size=128, semi=(0.35, 0.45, 0.35), bound=1.0):
    """dense [G,G,G] occupancy (x-major) of the head-like ellipsoid and the integer cell coordinates [G^3, 3]"""
    c = np.arange(grid_size, dtype=np.int32)
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    xyz = (coords.astype(np.float32) + 0.5) / grid_size * 2 * bound - bound
    inside = ((xyz / np.array(semi, dtype=np.float32)) ** 2).sum(1) <= 1.0
    return inside, coords


def ellipsoid_bitfield_device(device, grid_size=128, semi=(0.35, 0.45, 0.35), bound=1.0):
    """SURVEY 8d occupancy variant (i), built on the GPU with the product's own operators (raymarching.morton3D + packbits,
    the way update_extra_state does it, renderer.py:737-766) -> (bitfield uint8 [G^3/8] cuda, grid f32 [1, G^3] cuda)"""
    import torch

    from . import raymarching as R
    inside, coords = ellipsoid_grid(grid_size, semi, bound)
    idx = R.morton3D(torch.from_numpy(coords).to(device)).long()
    grid = torch.zeros(1, grid_size ** 3, dtype=torch.float32, device=device)
    grid[0, idx] = torch.from_numpy(inside.astype(np.float32)).to(device)
    return R.packbits(grid, 0.5), grid


def ones_bitfield(grid_size=128, cascade=1):
    """occupancy variant (ii): every cell occupied -- the dense deterministic workload of the headline"""
    return np.full(cascade * grid_size ** 3 // 8, 255, np.uint8)


class GenericHashgridNeRF:
    """BASELINE cfg2's workload: a generic (non-triplane) NeRF through the OPERATOR API only -- `get_encoder('hashgrid')` defaults (D 3,
    L 16, C 2, T 2^19, desired resolution 2048: encoding.py:6-8), SH(4) directions, two small bias-free MLPs on the MFMA Linear
    kernels (lzzx_nerf_amd.linear.MLP, the drop-in for network.py:73-94) -- rendered with the reference's inference loop as its callers
    write it (renderer.py:495-561: march_rays -> network -> composite_rays -> `rays_alive[rays_alive >= 0]`, n_step schedule), host
    sync per iteration included.  tests/test_gpu_cfg2_render.py checks the same arrangement against the CPU checker."""

    def __init__(self, device, seed=3, half_tables=False):
        import torch

        from .encoding import get_encoder
        from .linear import MLP
        self.enc, dim = get_encoder("hashgrid")
        self.sh, dim_sh = get_encoder("spherical_harmonics")
        g = torch.Generator().manual_seed(seed)
        self.enc = self.enc.to(device)
        self.enc.embeddings.data.copy_(torch.rand(self.enc.embeddings.shape, generator=g) * 2 - 1)
        self.half_tables = half_tables
        self.sigma_net, self.color_net = MLP(dim, 16, 64, 2).to(device), MLP(dim_sh + 15, 3, 64, 2).to(device)
        for m in (self.sigma_net, self.color_net):
            for lin in m.net:
                k = lin.weight.shape[1]
                lin.weight.data.copy_((torch.rand(lin.weight.shape, generator=g) * 2 - 1) / k ** 0.5)
        self.sigma_net.net[1].weight.data[0] *= 6.0   # sharper density: rays terminate, the schedule and the compaction are exercised

    def net(self, xyzs, dirs, bound):
        import torch
        if self.half_tables:       # what grid.py:28,38-39 does under autocast with an even level_dim: half tables, half features
            with torch.autocast("cuda", dtype=torch.float16):
                feat = self.enc(xyzs, bound=bound).float()
        else:
            feat = self.enc(xyzs, bound=bound)
        h = self.sigma_net(feat)
        sigma = torch.exp(h[:, 0])
        rgb = torch.sigmoid(self.color_net(torch.cat([self.sh(dirs), h[:, 1:]], -1)))
        return sigma, rgb

    def render(self, rays_o, rays_d, aabb, bitfield, bound=1.0, max_steps=128, T_thresh=1e-4, dt_gamma=1 / 256, min_near=0.05):
        """-> (image [N,3] blended on white, depth [N], weights_sum [N], marched sample rows, iterations)"""
        import torch

        from . import raymarching as R
        with torch.no_grad():
            N, dev = rays_o.shape[0], rays_o.device
            nears, fars = R.near_far_from_aabb(rays_o, rays_d, aabb, min_near)
            ws, dep, img = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
            alive = torch.arange(N, dtype=torch.int32, device=dev)
            t = nears.clone()
            step = rows = iters = 0
            while step < max_steps:
                n_alive = alive.shape[0]
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, dl = R.march_rays(n_alive, n_step, alive, t, rays_o, rays_d, bound, bitfield, 1, 128, nears, fars, 128, False, dt_gamma, max_steps)
                sigma, rgb = self.net(xyzs, dirs, bound)
                R.composite_rays(n_alive, n_step, alive, t, sigma, rgb, dl, ws, dep, img, T_thresh)
                alive = alive[alive >= 0]        # the reference's compaction: a device -> host sync per iteration (renderer.py:542)
                step += n_step
                rows += n_alive * n_step
                iters += 1
            return torch.clamp(img + (1 - ws)[:, None], 0, 1), dep, ws, rows, iters
