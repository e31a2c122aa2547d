"""ctypes bindings for the CPU checker (oracle/_build/liblzzx_oracle.so).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; nothing under lzzx_nerf_amd/ does.  Every function takes and returns numpy
arrays and mirrors one entry point of the reference's pybind modules
(/root/reference/{gridencoder,shencoder,freqencoder,raymarching}/src/*.h), with the allocation /
padding rules of the reference's Python wrappers restated next to the call.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblzzx_oracle.so")


def _stale():
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    inc = os.path.join(os.path.dirname(_HERE), "include")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")] + [os.path.join(inc, f) for f in os.listdir(inc)]
    return any(os.path.getmtime(f) > t for f in srcs)


def build(force=False):
    if force or _stale():
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def build_fast():
    """the second checker library: every product fused into the addition behind it that the compiler can reach, across statements too
    (oracle/Makefile `fast`) -- the other reading of nvcc's -fmad; tests/test_fma_contraction_bound.py"""
    so = os.path.join(_HERE, "_build", "liblzzx_oracle_fast.so")
    subprocess.check_call(["make", "-C", _HERE, "fast"], stdout=subprocess.DEVNULL)
    return so


class variant:
    """`with oracle.variant(path):` -- every checker call inside goes to another build of the library"""

    def __init__(self, so_path):
        self.so = so_path

    def __enter__(self):
        global _lib
        lib()
        self.prev = _lib
        _lib = C.CDLL(self.so)
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands each job a share of
    its host: 256 CPUs in the mask, a quota of 16)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, (q + p // 2) // p))
        except (OSError, ValueError):
            pass
    return n


def set_threads(n):
    lib().lzo_set_threads(C.c_int(int(n)))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


u32, f32c, i32c = C.c_uint32, C.c_float, C.c_int


# ----------------------------------------------------------------------------------------------
# gridencoder
# ----------------------------------------------------------------------------------------------
def grid_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners=False):
    """GridEncoder.__init__ table layout, gridencoder/grid.py:108-121"""
    off = np.zeros(num_levels + 1, dtype=np.int32)
    lib().lzo_grid_offsets(u32(input_dim), u32(num_levels), C.c_double(per_level_scale), u32(base_resolution),
                           u32(log2_hashmap_size), i32c(int(align_corners)), _p(off))
    return off


def grid_level_params(L, S, H):
    sc = np.zeros(L, dtype=np.float32)
    res = np.zeros(L, dtype=np.uint32)
    for l in range(L):
        s, r = C.c_float(), C.c_uint32()
        lib().lzo_grid_level_params(u32(l), f32c(S), u32(H), C.byref(s), C.byref(r))
        sc[l], res[l] = s.value, r.value
    return sc, res


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                        gridtype=0, align_corners=False):
    """_grid_encode.forward, gridencoder/grid.py:19-58.  Returns (outputs [B, L*C], dy_dx or None).
    embeddings float32 or float16 (the autocast branch grid.py:38-39 is the caller's choice)."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = embeddings.shape[1]
    S = np.float32(np.log2(per_level_scale))  # narrowed to float at the binding, gridencoder.h:12
    f16 = embeddings.dtype == np.float16
    emb = np.ascontiguousarray(embeddings)
    out = np.empty((L, B, Cc), dtype=emb.dtype)
    dy_dx = np.empty((B, L * D * Cc), dtype=emb.dtype) if calc_grad_inputs else None
    lib().lzo_grid_encode_forward(_p(inputs), _p(emb), _p(_i32(offsets)), _p(out), u32(B), u32(D), u32(Cc), u32(L),
                                  f32c(S), u32(base_resolution), _p(dy_dx), u32(gridtype), i32c(int(align_corners)),
                                  i32c(int(f16)))
    return np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(B, L * Cc), dy_dx  # grid.py:52


def grid_corner_indices(inputs, offsets, C_, per_level_scale, base_resolution, gridtype=0, align_corners=False):
    inputs = _f32(inputs)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    S = np.float32(np.log2(per_level_scale))
    out = np.empty((L, B, 1 << D), dtype=np.int32)
    lib().lzo_grid_corner_indices(_p(inputs), _p(_i32(offsets)), _p(out), u32(B), u32(D), u32(C_), u32(L), f32c(S),
                                  u32(base_resolution), u32(gridtype), i32c(int(align_corners)))
    return out


def grid_encode_backward(grad, inputs, embeddings_shape, offsets, per_level_scale, base_resolution, dy_dx=None,
                         gridtype=0, align_corners=False):
    """_grid_encode.backward, grid.py:60-84.  grad: [B, L*C] float32.  Returns (grad_embeddings, grad_inputs|None)"""
    inputs = _f32(inputs)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = embeddings_shape[1]
    S = np.float32(np.log2(per_level_scale))
    g = np.ascontiguousarray(_f32(grad).reshape(B, L, Cc).transpose(1, 0, 2))  # grid.py:70
    ge = np.zeros(embeddings_shape, dtype=np.float32)
    gi = np.zeros((B, D), dtype=np.float32) if dy_dx is not None else None
    lib().lzo_grid_encode_backward(_p(g), _p(inputs), _p(_i32(offsets)), _p(ge), u32(B), u32(D), u32(Cc), u32(L), f32c(S),
                                   u32(base_resolution), _p(None if dy_dx is None else _f32(dy_dx)), _p(gi), u32(gridtype),
                                   i32c(int(align_corners)))
    return ge, gi


def grid_encode_backward_f16(grad, inputs, embeddings_shape, offsets, per_level_scale, base_resolution, dy_dx=None,
                             gridtype=0, align_corners=False):
    """_grid_encode.backward with half tables (the autocast branch, grid.py:38-39 + gridencoder.cu:296-311).  grad: [B, L*C] float16.
    Returns dict(grad_embeddings half [sO, C] accumulated in (level, sample, corner) order, exact float64 sum of the same half terms,
    absum of their magnitudes, terms int32 count per entry, grad_inputs half [B, D] | None)."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = embeddings_shape[1]
    S = np.float32(np.log2(per_level_scale))
    g = np.ascontiguousarray(np.asarray(grad, dtype=np.float16).reshape(B, L, Cc).transpose(1, 0, 2))
    ge = np.zeros(embeddings_shape, dtype=np.float16)
    exact = np.zeros(embeddings_shape, dtype=np.float64)
    absum = np.zeros(embeddings_shape, dtype=np.float64)
    terms = np.zeros(embeddings_shape, dtype=np.int32)
    gi = np.zeros((B, D), dtype=np.float16) if dy_dx is not None else None
    dd = None if dy_dx is None else np.ascontiguousarray(np.asarray(dy_dx, dtype=np.float16))
    lib().lzo_grid_encode_backward_f16(_p(g), _p(inputs), _p(_i32(offsets)), _p(ge), _p(exact), _p(absum), _p(terms), u32(B), u32(D),
                                       u32(Cc), u32(L), f32c(S), u32(base_resolution), _p(dd), _p(gi), u32(gridtype),
                                       i32c(int(align_corners)))
    return dict(grad_embeddings=ge, exact=exact, absum=absum, terms=terms, grad_inputs=gi)


# ----------------------------------------------------------------------------------------------
# shencoder / freqencoder
# ----------------------------------------------------------------------------------------------
def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    inputs = _f32(inputs)
    B = inputs.shape[0]
    out = np.empty((B, degree * degree), dtype=np.float32)
    dy_dx = np.empty((B, 3 * degree * degree), dtype=np.float32) if calc_grad_inputs else None
    lib().lzo_sh_encode_forward(_p(inputs), _p(out), u32(B), u32(degree), _p(dy_dx))
    return out, dy_dx


def sh_encode_backward(grad, dy_dx, degree):
    grad = _f32(grad)
    B = grad.shape[0]
    gi = np.zeros((B, 3), dtype=np.float32)
    lib().lzo_sh_encode_backward(_p(grad), _p(_f32(dy_dx)), u32(B), u32(degree), _p(gi))
    return gi


def freq_encode_forward(inputs, degree):
    inputs = _f32(inputs)
    B, D = inputs.shape
    Cc = D + D * 2 * degree
    out = np.empty((B, Cc), dtype=np.float32)
    lib().lzo_freq_encode_forward(_p(inputs), u32(B), u32(D), u32(degree), u32(Cc), _p(out))
    return out


def freq_encode_backward(grad, outputs, input_dim, degree):
    grad = _f32(grad)
    B, Cc = grad.shape
    gi = np.zeros((B, input_dim), dtype=np.float32)
    lib().lzo_freq_encode_backward(_p(grad), _p(_f32(outputs)), u32(B), u32(input_dim), u32(degree), u32(Cc), _p(gi))
    return gi


# ----------------------------------------------------------------------------------------------
# linear / elementwise
# ----------------------------------------------------------------------------------------------
def fma(a, b, c):
    """elementwise float32 fused multiply-add"""
    a, b, c = np.broadcast_arrays(_f32(a), _f32(b), _f32(c))
    a, b, c = np.ascontiguousarray(a), np.ascontiguousarray(b), np.ascontiguousarray(c)
    y = np.empty_like(a)
    lib().lzo_vec_fma(_p(a), _p(b), _p(c), _p(y), C.c_size_t(a.size))
    return y


def linear_lanes(x, W=None):
    """y = x @ W.T in the lane-partial order of the fused head's VALU layers (lzo_linear_lanes); W None: sum of squares of x"""
    x = _f32(x)
    B, K = x.shape
    if W is None:
        N, ldw, wp = 1, 0, None
    else:
        W = _f32(W)
        N, ldw, wp = W.shape[0], K, W
        assert W.shape[1] == K
    y = np.empty((B, N), dtype=np.float32)
    lib().lzo_linear_lanes(_p(x), u32(K), _p(wp), u32(ldw), u32(K), u32(B), u32(N), _p(y), u32(N))
    return y


def linear(x, W, korder=None, relu=False):
    """y = x @ W.T as an fma chain in the given order (network.py:73-94; bias-free)."""
    x = _f32(x)
    W = _f32(W)
    B, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K
    if korder is not None:
        korder = _i32(korder)
        nk = korder.shape[0]
    else:
        nk = K
    y = np.empty((B, N), dtype=np.float32)
    lib().lzo_linear(_p(x), u32(K), _p(W), u32(K), _p(korder), u32(nk), u32(B), u32(N), i32c(int(relu)), _p(y), u32(N))
    return y


_OPS = {"exp": 0, "sigmoid": 1, "softplus": 2, "sin": 3, "log": 4}


def unary(op, x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().lzo_vec_unary(i32c(_OPS[op]), _p(x), _p(y), C.c_size_t(x.size))
    return y


# ----------------------------------------------------------------------------------------------
# raymarching
# ----------------------------------------------------------------------------------------------
def get_rays(pose, intrinsics, H, W):
    """full image, one pose [4,4] -> rays_o, rays_d [H*W, 3]"""
    r = get_rays_batched(_f32(pose).reshape(1, 4, 4), intrinsics, H, W)
    return r["rays_o"][0], r["rays_d"][0]


def get_rays_batched(poses, intrinsics, H, W, inds=None):
    """utils.py:226-312 for given pixel indices: poses [B,4,4], inds int64 [N] or None (all pixels) -> dict(i, j, rays_o, rays_d)"""
    poses = _f32(poses).reshape(-1, 4, 4)
    B = poses.shape[0]
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    if inds is not None:
        inds = np.ascontiguousarray(inds, dtype=np.int64).reshape(-1)
    N = H * W if inds is None else inds.shape[0]
    ro, rd = np.empty((B, N, 3), dtype=np.float32), np.empty((B, N, 3), dtype=np.float32)
    oi, oj = np.empty((B, N), dtype=np.float32), np.empty((B, N), dtype=np.float32)
    lib().lzo_get_rays(_p(poses), f32c(fx), f32c(fy), f32c(cx), f32c(cy), u32(H), u32(W), u32(B), u32(N),
                       None if inds is None else _p(inds), _p(ro), _p(rd), _p(oi), _p(oj))
    return dict(i=oi, j=oj, rays_o=ro, rays_d=rd)


def bg_coords(H, W):
    out = np.empty((H * W, 2), dtype=np.float32)
    lib().lzo_bg_coords(u32(H), u32(W), _p(out))
    return out


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o = _f32(rays_o).reshape(-1, 3)
    rays_d = _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    nears = np.empty(N, dtype=np.float32)
    fars = np.empty(N, dtype=np.float32)
    lib().lzo_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(_f32(aabb)), u32(N), f32c(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o = _f32(rays_o).reshape(-1, 3)
    rays_d = _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), dtype=np.float32)
    lib().lzo_sph_from_ray(_p(rays_o), _p(rays_d), f32c(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _i32(coords)
    N = coords.shape[0]
    out = np.empty(N, dtype=np.int32)
    lib().lzo_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices)
    N = indices.shape[0]
    out = np.empty((N, 3), dtype=np.int32)
    lib().lzo_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh):
    grid = _f32(grid)
    N = grid.size // 8
    out = np.empty(N, dtype=np.uint8)
    lib().lzo_packbits(_p(grid), u32(N), f32c(thresh), _p(out))
    return out


def morton3D_dilation(grid):
    grid = _f32(grid)
    Cc, H3 = grid.shape
    H = int(round(H3 ** (1.0 / 3.0)))
    out = np.empty_like(grid)
    lib().lzo_morton3D_dilation(_p(grid), u32(Cc), u32(H), _p(out))
    return out


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C_, H, nears, fars, step_counter=None, mean_count=-1,
                     noises=None, align=-1, force_all_rays=False, dt_gamma=0.0, max_steps=1024):
    """_march_rays_train.forward, raymarching/raymarching.py:186-260 (noises passed in instead of torch.rand)."""
    rays_o = _f32(rays_o).reshape(-1, 3)
    rays_d = _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    xyzs = np.zeros((M, 3), dtype=np.float32)
    dirs = np.zeros((M, 3), dtype=np.float32)
    deltas = np.zeros((M, 2), dtype=np.float32)
    rays = np.empty((N, 3), dtype=np.int32)
    if step_counter is None:
        step_counter = np.zeros(2, dtype=np.int32)
    if noises is None:
        noises = np.zeros(N, dtype=np.float32)
    lib().lzo_march_rays_train(_p(rays_o), _p(rays_d), _p(np.ascontiguousarray(density_bitfield, dtype=np.uint8)),
                               f32c(bound), f32c(dt_gamma), u32(max_steps), u32(N), u32(C_), u32(H), u32(M),
                               _p(_f32(nears)), _p(_f32(fars)), _p(xyzs), _p(dirs), _p(deltas), _p(rays),
                               _p(step_counter), _p(_f32(noises)))
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0])
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas):
    N = rays.shape[0]
    M = grad_xyzs.shape[0]
    go = np.zeros((N, 3), dtype=np.float32)
    gd = np.zeros((N, 3), dtype=np.float32)
    lib().lzo_march_rays_train_backward(_p(_f32(grad_xyzs)), _p(_f32(grad_dirs)), _p(_i32(rays)), _p(_f32(deltas)),
                                        u32(N), u32(M), _p(go), _p(gd))
    return go, gd


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C_, H, nears, fars,
               align=-1, noises=None, dt_gamma=0.0, max_steps=1024):
    """_march_rays.forward, raymarching.py:347-396"""
    rays_o = _f32(rays_o).reshape(-1, 3)
    rays_d = _f32(rays_d).reshape(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = np.zeros((M, 3), dtype=np.float32)
    dirs = np.zeros((M, 3), dtype=np.float32)
    deltas = np.zeros((M, 2), dtype=np.float32)
    if noises is None:
        noises = np.zeros(n_alive, dtype=np.float32)
    lib().lzo_march_rays(u32(n_alive), u32(n_step), _p(_i32(rays_alive)), _p(_f32(rays_t)), _p(rays_o), _p(rays_d),
                         f32c(bound), f32c(dt_gamma), u32(max_steps), u32(C_), u32(H),
                         _p(np.ascontiguousarray(density_bitfield, dtype=np.uint8)), _p(_f32(nears)), _p(_f32(fars)),
                         _p(xyzs), _p(dirs), _p(deltas), _p(_f32(noises)))
    return xyzs, dirs, deltas


# variant -> (n_amb, amb_weighted, has_unc)
VARIANTS = {
    "plain": (0, 0, 0),          # composite_rays (inference only)
    "ambient": (1, 0, 0),        # composite_rays_train / composite_rays_ambient
    "sigma": (1, 1, 0),          # composite_rays_train_sigma / composite_rays_ambient_sigma
    "uncertainty": (1, 0, 1),    # composite_rays_train_uncertainty / composite_rays_uncertainty
    "triplane": (2, 0, 1),       # composite_rays_train_triplane / composite_rays_triplane
}


def composite_rays_train_forward(variant, sigmas, rgbs, deltas, rays, amb0=None, amb1=None, unc=None, T_thresh=1e-4):
    na, aw, hu = VARIANTS[variant]
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, d = np.empty(N, np.float32), np.empty(N, np.float32)
    img = np.empty((N, 3), np.float32)
    a0s = np.empty(N, np.float32) if na > 0 else None
    a1s = np.empty(N, np.float32) if na > 1 else None
    us = np.empty(N, np.float32) if hu else None
    lib().lzo_composite_rays_train_forward(
        _p(sigmas), _p(rgbs), _p(None if amb0 is None else _f32(amb0)), _p(None if amb1 is None else _f32(amb1)),
        _p(None if unc is None else _f32(unc)), _p(deltas), _p(rays), u32(M), u32(N), f32c(T_thresh),
        i32c(na), i32c(aw), i32c(hu), _p(ws), _p(a0s), _p(a1s), _p(us), _p(d), _p(img))
    return dict(weights_sum=ws, amb0_sum=a0s, amb1_sum=a1s, unc_sum=us, depth=d, image=img)


def composite_rays_train_backward(variant, grads, sigmas, rgbs, deltas, rays, fwd, amb0=None, amb1=None, unc=None,
                                  T_thresh=1e-4):
    """grads: dict with grad_weights_sum, grad_image, and (per variant) grad_amb0_sum, grad_amb1_sum, grad_unc_sum"""
    na, aw, hu = VARIANTS[variant]
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    gs = np.zeros(M, np.float32)
    gr = np.zeros((M, 3), np.float32)
    ga0 = np.zeros(M, np.float32) if na > 0 else None
    ga1 = np.zeros(M, np.float32) if na > 1 else None
    gu = np.zeros(M, np.float32) if hu else None
    g = {k: (None if v is None else _f32(v)) for k, v in grads.items()}
    lib().lzo_composite_rays_train_backward(
        _p(g["grad_weights_sum"]), _p(g.get("grad_amb0_sum")), _p(g.get("grad_amb1_sum")), _p(g.get("grad_unc_sum")),
        _p(g["grad_image"]), _p(sigmas), _p(rgbs), _p(None if amb0 is None else _f32(amb0)),
        _p(None if amb1 is None else _f32(amb1)), _p(None if unc is None else _f32(unc)), _p(deltas), _p(rays),
        _p(fwd["weights_sum"]), _p(fwd.get("amb0_sum")), _p(fwd.get("unc_sum")), _p(fwd["image"]),
        u32(M), u32(N), f32c(T_thresh), i32c(na), i32c(aw), i32c(hu), _p(gs), _p(gr), _p(ga0), _p(ga1), _p(gu))
    return dict(grad_sigmas=gs, grad_rgbs=gr, grad_amb0=ga0, grad_amb1=ga1, grad_unc=gu)


def composite_rays(variant, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                   amb0=None, amb1=None, unc=None, amb0_sum=None, amb1_sum=None, unc_sum=None, T_thresh=1e-2):
    """In place on rays_alive, rays_t and the accumulators (all must be contiguous numpy arrays of the right dtype)."""
    na, aw, hu = VARIANTS[variant]
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32),
                  (image, np.float32)):
        assert a.dtype == dt and a.flags.c_contiguous
    lib().lzo_composite_rays(
        u32(n_alive), u32(n_step), f32c(T_thresh), _p(rays_alive), _p(rays_t), _p(_f32(sigmas)), _p(_f32(rgbs)),
        _p(_f32(deltas)), _p(None if amb0 is None else _f32(amb0)), _p(None if amb1 is None else _f32(amb1)),
        _p(None if unc is None else _f32(unc)), i32c(na), i32c(aw), i32c(hu), _p(weights_sum), _p(depth), _p(image),
        _p(amb0_sum), _p(amb1_sum), _p(unc_sum))
