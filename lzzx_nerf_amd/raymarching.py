"""Ray marching / compositing operators -- the API of the reference's `raymarching` package
(/root/reference/raymarching/raymarching.py:18-671: 17 autograd Functions) on the gfx950 kernels
(csrc/lz_raymarch.hip).  Positional signatures, defaults, output allocation/padding rules and in-place
semantics are the reference's; every floating-point wrapper casts to float32 like `custom_fwd(cast_inputs=...)`.

The five compositing variants share three kernels selected by (n_amb, amb_weighted, has_unc):
plain (0,0,0), ambient (1,0,0), sigma (1,1,0), uncertainty (1,0,1), triplane (2,0,1).
"""
import numpy as np
import torch
from torch.autograd import Function

from ._util import call, ptr, stream

_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


def _cuda(t):
    return t if t.is_cuda else t.cuda()


# ----------------------------------------
# utils
# ----------------------------------------
class _near_far_from_aabb(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        aabb = aabb.contiguous()   # converted copies stay bound to a name until the launch is enqueued
        call("lz_near_far_from_aabb", ptr(rays_o), ptr(rays_d), ptr(aabb), N, float(min_near), ptr(nears), ptr(fars), stream())
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, radius):
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        call("lz_sph_from_ray", ptr(rays_o), ptr(rays_d), float(radius), N, ptr(coords), stream())
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        coords = _cuda(coords).int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        call("lz_morton3D", ptr(coords), N, ptr(indices), stream())
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        indices = _cuda(indices).int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        call("lz_morton3D_invert", ptr(indices), N, ptr(coords), stream())
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, grid, thresh, bitfield=None):
        grid = _cuda(grid).contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        call("lz_packbits", ptr(grid), N, float(thresh), ptr(bitfield), stream())
        torch.autograd.graph.increment_version(bitfield)   # a caller-supplied bitfield was written through its raw pointer
        return bitfield


packbits = _packbits.apply


class _morton3D_dilation(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, grid):
        grid = _cuda(grid).contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        H = int(np.cbrt(H3))
        if H * H * H != H3:  # np.cbrt of a perfect cube can land one below (e.g. 127.99999)
            H = int(round(H3 ** (1.0 / 3.0)))
        out = torch.empty_like(grid)
        call("lz_morton3D_dilation", ptr(grid), C, H, ptr(out), stream())
        return out


morton3D_dilation = _morton3D_dilation.apply


# ----------------------------------------
# train functions
# ----------------------------------------
# ---- sample-row layout of the training path (round 5) ----------------------------------------------------------------------------
# "ray"  : the reference's -- a ray's samples in consecutive rows (raymarching.cu:446-517), rays in ray-id order;
# "step" : step-major groups of 64 rays taken in a locality order (include/lzzx_nerf_hip.h: lz_march_rays_train_grouped) -- the rows a
#          wave of the head / encoders / compositing works on are 64 NEIGHBOURING rays at the same step instead of 64 consecutive
#          samples of one ray.  Nothing downstream of the reference's run_cuda depends on which rows a ray got (its own atomics leave
#          that open); the two operators that do -- compositing and the march's backward -- read the layout off the `rays` tensor
#          march_rays_train returned (attribute `lz_layout`), or take the module default when handed a copy of it.
_TRAIN_LAYOUT = "ray"
_LAYOUTS = {"ray": 0, "step": 1}


def set_train_layout(layout):
    """module default for march_rays_train(layout=None) and for compositing a `rays` tensor that carries no layout tag; returns the previous one"""
    global _TRAIN_LAYOUT
    if layout not in _LAYOUTS:
        raise ValueError(f"layout must be one of {sorted(_LAYOUTS)}, got {layout!r}")
    prev, _TRAIN_LAYOUT = _TRAIN_LAYOUT, layout
    return prev


def train_layout():
    return _TRAIN_LAYOUT


def _layout_of(rays):
    return _LAYOUTS[getattr(rays, "lz_layout", None) or _TRAIN_LAYOUT]


def ray_order(rays_o, rays_d, bound):
    """the locality order of the step-major layout: int32 [N] permutation, neighbouring pixels next to each other (stable sort of
    lz_ray_sort_keys: direction by octahedral Morton code, cameras apart)"""
    rays_o = _cuda(rays_o).contiguous().view(-1, 3).float()
    rays_d = _cuda(rays_d).contiguous().view(-1, 3).float()
    N = rays_o.shape[0]
    keys = torch.empty(N, dtype=torch.int32, device=rays_o.device)
    call("lz_ray_sort_keys", ptr(rays_o), ptr(rays_d), N, float(bound), ptr(keys), stream())
    return torch.sort(keys, stable=True).indices.to(torch.int32)


class _march_rays_train(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, layout="ray", order=None):
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        density_bitfield = _cuda(density_bitfield).contiguous()
        N = rays_o.shape[0]
        M = N * max_steps
        # running-average cap on the number of points (raymarching.py:223-228)
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        dev = rays_o.device
        # (the reference fills these with torch.zeros, raymarching.py:246-248; lz_march_rays_train itself zeroes every row it does not write)
        xyzs = torch.empty(M, 3, dtype=rays_o.dtype, device=dev) if N > 0 else torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
        dirs = torch.empty(M, 3, dtype=rays_o.dtype, device=dev) if N > 0 else torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
        deltas = torch.empty(M, 2, dtype=rays_o.dtype, device=dev) if N > 0 else torch.zeros(M, 2, dtype=rays_o.dtype, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, dtype=rays_o.dtype, device=dev) if perturb else torch.zeros(N, dtype=rays_o.dtype, device=dev)
        workspace = torch.empty(N + 2, dtype=torch.int32, device=dev)
        nears, fars = nears.contiguous(), fars.contiguous()   # two live names: a freed temporary's block could be handed to the next one
        ctx.layout = _LAYOUTS[layout]
        auto_order = None
        if ctx.layout == 1:
            if order is None and N > 0:
                order = ray_order(rays_o, rays_d, bound)
                auto_order = order
            elif order is False:
                order = None                                  # groups of 64 in ray-id order
            if order is not None:
                given = order
                order = _cuda(order).to(torch.int32).contiguous()
                if order.numel() != N:
                    raise ValueError(f"march_rays_train: order has {order.numel()} entries for {N} rays")
                if given is not auto_order and N > 0:
                    # a caller's own order is read by the kernels as ray ids: anything but a permutation of 0..N-1 reads out of bounds or
                    # writes two rays into one row range.  One sort + a host sync, paid only by callers who bring their own order.
                    if not bool((torch.sort(order.long()).values == torch.arange(N, device=order.device)).all()):
                        raise ValueError("march_rays_train: order must be a permutation of 0..N-1")
            call("lz_march_rays_train_grouped", ptr(rays_o), ptr(rays_d), ptr(density_bitfield), float(bound), float(dt_gamma), int(max_steps), N,
                 int(C), int(H), M, ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays),
                 ptr(step_counter), ptr(noises), ptr(order), ptr(workspace), stream())
        else:
            call("lz_march_rays_train", ptr(rays_o), ptr(rays_d), ptr(density_bitfield), float(bound), float(dt_gamma), int(max_steps), N,
                 int(C), int(H), M, ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays),
                 ptr(step_counter), ptr(noises), ptr(workspace), stream())
        if force_all_rays or mean_count <= 0:
            m = step_counter[0].item()  # D2H copy, as in the reference (raymarching.py:249)
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        ctx.save_for_backward(rays, deltas)
        return xyzs, dirs, deltas, rays

    @staticmethod
    @_bwd
    def backward(ctx, grad_xyzs, grad_dirs, grad_deltas, grad_rays):
        rays, deltas = ctx.saved_tensors
        N, M = rays.shape[0], grad_xyzs.shape[0]
        grad_rays_o = torch.zeros(N, 3, device=rays.device)
        grad_rays_d = torch.zeros(N, 3, device=rays.device)
        gx, gd, deltas = grad_xyzs.float().contiguous(), grad_dirs.float().contiguous(), deltas.contiguous()
        call("lz_march_rays_train_backward_grouped" if ctx.layout == 1 else "lz_march_rays_train_backward", ptr(gx), ptr(gd), ptr(rays), ptr(deltas),
             N, M, ptr(grad_rays_o), ptr(grad_rays_d), stream())
        return (grad_rays_o, grad_rays_d) + (None,) * 15


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                     align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, layout=None, order=None):
    """raymarching.py:186-280, plus `layout` ("ray" | "step" | None = the module default, see set_train_layout) and, for "step", `order`
    (an int32 permutation of the rays; None = ray_order(rays_o, rays_d, bound); False = ray-id order).  The returned `rays` carries the
    layout (`rays.lz_layout`) for compositing."""
    layout = layout or _TRAIN_LAYOUT
    if layout not in _LAYOUTS:
        raise ValueError(f"layout must be one of {sorted(_LAYOUTS)}, got {layout!r}")
    out = _march_rays_train.apply(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter, mean_count, perturb, align,
                                  force_all_rays, dt_gamma, max_steps, layout, order)
    out[3].lz_layout = layout
    return out


def _composite_train_fwd(variant, sigmas, rgbs, amb0, amb1, unc, deltas, rays, T_thresh, layout=0):
    na, aw, hu = variant
    M, N = sigmas.shape[0], rays.shape[0]
    kw = dict(dtype=sigmas.dtype, device=sigmas.device)
    weights_sum, depth, image = torch.empty(N, **kw), torch.empty(N, **kw), torch.empty(N, 3, **kw)
    a0s = torch.empty(N, **kw) if na > 0 else None
    a1s = torch.empty(N, **kw) if na > 1 else None
    us = torch.empty(N, **kw) if hu else None
    call("lz_composite_train_forward_v", ptr(sigmas), ptr(rgbs), ptr(amb0), ptr(amb1), ptr(unc), ptr(deltas), ptr(rays), M, N,
         float(T_thresh), na, aw, hu, int(layout), ptr(weights_sum), ptr(a0s), ptr(a1s), ptr(us), ptr(depth), ptr(image), stream())
    return weights_sum, a0s, a1s, us, depth, image


def _composite_train_bwd(variant, g_ws, g_a0, g_a1, g_u, g_img, sigmas, rgbs, amb0, amb1, unc, deltas, rays, weights_sum, a0s, us,
                         image, T_thresh, layout=0):
    na, aw, hu = variant
    M, N = sigmas.shape[0], rays.shape[0]
    # ray-major rows: pre-zeroed like the reference's wrapper (raymarching.py:332-334, 649-653); the step-major kernel writes every row itself
    fresh = torch.empty_like if (layout == 1 and N > 0) else torch.zeros_like
    grad_sigmas, grad_rgbs = fresh(sigmas), fresh(rgbs)
    ga0 = fresh(amb0) if na > 0 else None
    ga1 = fresh(amb1) if na > 1 else None
    gu = fresh(unc) if hu else None
    call("lz_composite_train_backward_v", ptr(g_ws), ptr(g_a0), ptr(g_a1), ptr(g_u), ptr(g_img), ptr(sigmas), ptr(rgbs), ptr(amb0),
         ptr(amb1), ptr(unc), ptr(deltas), ptr(rays), ptr(weights_sum), ptr(a0s), ptr(us), ptr(image), M, N, float(T_thresh), na, aw, hu,
         int(layout), ptr(grad_sigmas), ptr(grad_rgbs), ptr(ga0), ptr(ga1), ptr(gu), stream())
    return grad_sigmas, grad_rgbs, ga0, ga1, gu


def _make_train_1amb(variant):
    """composite_rays_train (ambient unweighted) / composite_rays_train_sigma (ambient weighted): raymarching.py:283-341, 442-500"""

    class _Fn(Function):
        @staticmethod
        @_fwd32
        def forward(ctx, sigmas, rgbs, ambient, deltas, rays, T_thresh=1e-4):
            sigmas, rgbs, ambient = sigmas.contiguous(), rgbs.contiguous(), ambient.contiguous()
            deltas = deltas.contiguous()
            ctx.layout = _layout_of(rays)
            ws, a0s, _, _, depth, image = _composite_train_fwd(variant, sigmas, rgbs, ambient, None, None, deltas, rays, T_thresh, ctx.layout)
            ctx.save_for_backward(sigmas, rgbs, ambient, deltas, rays, ws, a0s, depth, image)
            ctx.T_thresh = T_thresh
            return ws, a0s, depth, image

        @staticmethod
        @_bwd
        def backward(ctx, grad_weights_sum, grad_ambient_sum, grad_depth, grad_image):
            # grad_depth is not propagated (raymarching.py:323)
            sigmas, rgbs, ambient, deltas, rays, ws, a0s, depth, image = ctx.saved_tensors
            gs, gr, ga, _, _ = _composite_train_bwd(variant, grad_weights_sum.contiguous(), grad_ambient_sum.contiguous(), None, None,
                                                   grad_image.contiguous(), sigmas, rgbs, ambient, None, None, deltas, rays, ws, a0s,
                                                   None, image, ctx.T_thresh, ctx.layout)
            return gs, gr, ga, None, None, None

    return _Fn


_composite_rays_train = _make_train_1amb((1, 0, 0))
composite_rays_train = _composite_rays_train.apply
_composite_rays_train_sigma = _make_train_1amb((1, 1, 0))
composite_rays_train_sigma = _composite_rays_train_sigma.apply


class _composite_rays_train_uncertainty(Function):  # raymarching.py:516-578
    @staticmethod
    @_fwd32
    def forward(ctx, sigmas, rgbs, ambient, uncertainty, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, ambient, uncertainty = sigmas.contiguous(), rgbs.contiguous(), ambient.contiguous(), uncertainty.contiguous()
        deltas = deltas.contiguous()
        ctx.layout = _layout_of(rays)
        ws, a0s, _, us, depth, image = _composite_train_fwd((1, 0, 1), sigmas, rgbs, ambient, None, uncertainty, deltas, rays, T_thresh, ctx.layout)
        ctx.save_for_backward(sigmas, rgbs, ambient, uncertainty, deltas, rays, ws, a0s, us, depth, image)
        ctx.T_thresh = T_thresh
        return ws, a0s, us, depth, image

    @staticmethod
    @_bwd
    def backward(ctx, grad_weights_sum, grad_ambient_sum, grad_uncertainty_sum, grad_depth, grad_image):
        sigmas, rgbs, ambient, uncertainty, deltas, rays, ws, a0s, us, depth, image = ctx.saved_tensors
        gs, gr, ga, _, gu = _composite_train_bwd((1, 0, 1), grad_weights_sum.contiguous(), grad_ambient_sum.contiguous(), None,
                                                 grad_uncertainty_sum.contiguous(), grad_image.contiguous(), sigmas, rgbs, ambient, None,
                                                 uncertainty, deltas, rays, ws, a0s, us, image, ctx.T_thresh, ctx.layout)
        return gs, gr, ga, gu, None, None, None


composite_rays_train_uncertainty = _composite_rays_train_uncertainty.apply


class _composite_rays_train_triplane(Function):  # raymarching.py:594-660
    @staticmethod
    @_fwd32
    def forward(ctx, sigmas, rgbs, amb_aud, amb_eye, uncertainty, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs = sigmas.contiguous(), rgbs.contiguous()
        amb_aud, amb_eye, uncertainty = amb_aud.contiguous(), amb_eye.contiguous(), uncertainty.contiguous()
        deltas = deltas.contiguous()
        ctx.layout = _layout_of(rays)
        ws, a0s, a1s, us, depth, image = _composite_train_fwd((2, 0, 1), sigmas, rgbs, amb_aud, amb_eye, uncertainty, deltas, rays, T_thresh, ctx.layout)
        ctx.save_for_backward(sigmas, rgbs, amb_aud, amb_eye, uncertainty, deltas, rays, ws, a0s, a1s, us, depth, image)
        ctx.T_thresh = T_thresh
        return ws, a0s, a1s, us, depth, image

    @staticmethod
    @_bwd
    def backward(ctx, grad_weights_sum, grad_amb_aud_sum, grad_amb_eye_sum, grad_uncertainty_sum, grad_depth, grad_image):
        sigmas, rgbs, amb_aud, amb_eye, uncertainty, deltas, rays, ws, a0s, a1s, us, depth, image = ctx.saved_tensors
        gs, gr, ga0, ga1, gu = _composite_train_bwd((2, 0, 1), grad_weights_sum.contiguous(), grad_amb_aud_sum.contiguous(),
                                                    grad_amb_eye_sum.contiguous(), grad_uncertainty_sum.contiguous(),
                                                    grad_image.contiguous(), sigmas, rgbs, amb_aud, amb_eye, uncertainty, deltas, rays,
                                                    ws, a0s, us, image, ctx.T_thresh, ctx.layout)
        return gs, gr, ga0, ga1, gu, None, None, None


composite_rays_train_triplane = _composite_rays_train_triplane.apply


# ----------------------------------------
# infer functions
# ----------------------------------------
class _march_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
                perturb=False, dt_gamma=0, max_steps=1024):
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)  # adds a full `align` when already divisible (raymarching.py:381-382)
        dev = rays_o.device
        xyzs = torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
        dirs = torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
        deltas = torch.zeros(M, 2, dtype=rays_o.dtype, device=dev)
        noises = torch.rand(n_alive, dtype=rays_o.dtype, device=dev) if perturb else None
        call("lz_march_rays", int(n_alive), int(n_step), ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), float(bound),
             float(dt_gamma), int(max_steps), int(C), int(H), ptr(density_bitfield), ptr(near), ptr(far), ptr(xyzs), ptr(dirs),
             ptr(deltas), ptr(noises), stream())
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


def _composite_infer(variant, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, amb0, amb1, unc, weights_sum, depth, image,
                     a0s, a1s, us, T_thresh):
    na, aw, hu = variant
    cont = lambda t: None if t is None else t.contiguous()
    sigmas, rgbs, deltas, amb0, amb1, unc = [cont(t) for t in (sigmas, rgbs, deltas, amb0, amb1, unc)]   # all alive until the launch
    call("lz_composite_rays_v", int(n_alive), int(n_step), float(T_thresh), ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(rgbs),
         ptr(deltas), ptr(amb0), ptr(amb1), ptr(unc), na, aw, hu, ptr(weights_sum), ptr(depth), ptr(image),
         ptr(a0s), ptr(a1s), ptr(us), stream())


class _composite_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        _composite_infer((0, 0, 0), n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, None, None, None, weights_sum, depth, image,
                         None, None, None, T_thresh)
        return tuple()


composite_rays = _composite_rays.apply


class _composite_rays_ambient(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, ambients, weights_sum, depth, image, ambient_sum,
                T_thresh=1e-2):
        _composite_infer((1, 0, 0), n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, ambients, None, None, weights_sum, depth,
                         image, ambient_sum, None, None, T_thresh)
        return tuple()


composite_rays_ambient = _composite_rays_ambient.apply


class _composite_rays_ambient_sigma(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, ambients, weights_sum, depth, image, ambient_sum,
                T_thresh=1e-2):
        _composite_infer((1, 1, 0), n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, ambients, None, None, weights_sum, depth,
                         image, ambient_sum, None, None, T_thresh)
        return tuple()


composite_rays_ambient_sigma = _composite_rays_ambient_sigma.apply


class _composite_rays_uncertainty(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, ambients, uncertainties, weights_sum, depth, image,
                ambient_sum, uncertainty_sum, T_thresh=1e-2):
        _composite_infer((1, 0, 1), n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, ambients, None, uncertainties, weights_sum,
                         depth, image, ambient_sum, None, uncertainty_sum, T_thresh)
        return tuple()


composite_rays_uncertainty = _composite_rays_uncertainty.apply


class _composite_rays_triplane(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, ambs_aud, ambs_eye, uncertainties, weights_sum, depth,
                image, amb_aud_sum, amb_eye_sum, uncertainty_sum, T_thresh=1e-2):
        # test-mode uncertainty arrives over-sized ([M,36,1], network.py:243-249,277-280); only its first M values are read,
        # all of them ln 2 -- exactly what the reference's kernel does with that buffer.
        _composite_infer((2, 0, 1), n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, ambs_aud, ambs_eye, uncertainties,
                         weights_sum, depth, image, amb_aud_sum, amb_eye_sum, uncertainty_sum, T_thresh)
        return tuple()


composite_rays_triplane = _composite_rays_triplane.apply
