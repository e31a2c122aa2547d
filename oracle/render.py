"""CPU checker for the render loops -- TEST INFRASTRUCTURE ONLY (also bench.py's `cpu_baseline` leg).

The reference has no CPU renderer (train.py:152 forces cuda_ray; renderer.py:952-954 raises otherwise),
so this file arranges the checker's kernels exactly like the reference's loops:
  * inference: NeRFRenderer.run_cuda_for_inference   /root/reference/nerf_triplane/renderer.py:436-561
  * training forward: NeRFRenderer.run_cuda, training branch   renderer.py:279-304, 380-385
PINNED (round 4): tests/golden/reference_loops.npz holds what the reference's OWN loops and raymarching wrappers produce when they are
run unmodified on the checker's kernels (tests/golden/make_golden_loops.py); tests/test_golden_loops.py holds these two functions to
it -- iteration schedule, buffer sizes, per-ray counts exactly; images bit for bit with the reference's MLP arrangement.
"""
import numpy as np

from . import oracle as O
from .head import head_forward

F32 = np.float32


def default_aabb(bound):
    """renderer.py:110 -- y extent halved"""
    return np.array([-bound, -bound / 2, -bound, bound, bound / 2, bound], dtype=F32)


def render_inference(spec, P, rays_o, rays_d, bitfield, enc_a, ind_code, eye, cascade=1, grid_size=128, aabb=None,
                     min_near=0.05, dt_gamma=1.0 / 256, max_steps=16, T_thresh=1e-4, bg_color=1.0, stats=None,
                     budget_factor=1, n_step_cap=8, head=None, noises=None, testing=True):
    """head: the per-sample network, default the bit-pinned checker `head_forward`; bench.py's cpu_baseline passes
    `head_forward_torch` (the reference's torch-CPU MLP arrangement).  noises [N]: `perturb` -- handed to march_rays on the first
    iteration only, like renderer.py:521 (`perturb if step == 0 else False`).  testing: NeRFNetwork.testing -- the test path sets it
    around render (TrainerUtil.py:432-436: uncertainty = ln 2), the evaluation path does not (:390-394: unc_net runs)"""
    if head is None:
        head = head_forward
    rays_o = np.ascontiguousarray(rays_o, dtype=F32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, dtype=F32).reshape(-1, 3)
    N = rays_o.shape[0]
    bound = spec.bound
    if aabb is None:
        aabb = default_aabb(bound)
    nears, fars = O.near_far_from_aabb(rays_o, rays_d, aabb, min_near)            # renderer.py:442
    weights_sum = np.zeros(N, F32)
    depth = np.zeros(N, F32)
    image = np.zeros((N, 3), F32)
    amb_aud_sum = np.zeros(N, F32)
    amb_eye_sum = np.zeros(N, F32)
    unc_sum = np.zeros(N, F32)
    rays_alive = np.arange(N, dtype=np.int32)                                      # renderer.py:496
    rays_t = nears.copy()
    counts = np.zeros(N, np.int64)
    schedule = []
    step = 0
    while step < max_steps:                                                        # renderer.py:503
        n_alive = rays_alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(budget_factor * N // n_alive, n_step_cap), 1)           # renderer.py:513 (factor 1, cap 8)
        xyzs, dirs, deltas = O.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield,
                                          cascade, grid_size, nears, fars, 128, noises if step == 0 else None, dt_gamma, max_steps)
        sigmas, rgbs, amb_aud, amb_eye, unc = head(spec, P, xyzs, dirs, enc_a, ind_code, eye, testing=testing)
        if amb_eye is None:
            amb_eye = np.zeros_like(amb_aud)
        valid = (deltas[: n_alive * n_step, 0] != 0).reshape(n_alive, n_step).sum(1)
        np.add.at(counts, rays_alive, valid)
        schedule.append((n_alive, n_step))
        O.composite_rays("triplane", n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth,
                         image, amb0=amb_aud, amb1=amb_eye, unc=unc, amb0_sum=amb_aud_sum, amb1_sum=amb_eye_sum,
                         unc_sum=unc_sum, T_thresh=T_thresh)
        rays_alive = np.ascontiguousarray(rays_alive[rays_alive >= 0])             # renderer.py:542
        step += n_step
    final = image + (F32(1) - weights_sum)[:, None] * F32(bg_color)                # renderer.py:559-561
    final = np.clip(final, F32(0), F32(1)).astype(F32)
    if stats is not None:
        stats.update(samples_per_ray=counts, schedule=schedule, nears=nears, fars=fars)
    return dict(image=final, image_raw=image, weights_sum=weights_sum, depth=depth, amb_aud_sum=amb_aud_sum,
                amb_eye_sum=amb_eye_sum, uncertainty_sum=unc_sum)


def render_train_forward(spec, P, rays_o, rays_d, bitfield, enc_a, ind_code, eye, cascade=1, grid_size=128, aabb=None,
                         min_near=0.05, dt_gamma=1.0 / 256, max_steps=16, T_thresh=1e-4, bg_color=1.0, noises=None,
                         mean_count=-1, force_all_rays=False, unc_loss=True, head=None):
    """head: None = the bit-pinned checker head_forward; head_forward_torch = the reference's torch-CPU MLP arrangement"""
    rays_o = np.ascontiguousarray(rays_o, dtype=F32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, dtype=F32).reshape(-1, 3)
    bound = spec.bound
    if aabb is None:
        aabb = default_aabb(bound)
    nears, fars = O.near_far_from_aabb(rays_o, rays_d, aabb, min_near)
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = O.march_rays_train(rays_o, rays_d, bound, bitfield, cascade, grid_size, nears, fars,
                                                  counter, mean_count, noises, 128, force_all_rays, dt_gamma, max_steps)
    if head is None:
        sigmas, rgbs, amb_aud, amb_eye, unc = head_forward(spec, P, xyzs, dirs, enc_a, ind_code, eye, testing=False, unc_loss=unc_loss)
    else:
        sigmas, rgbs, amb_aud, amb_eye, unc = head(spec, P, xyzs, dirs, enc_a, ind_code, eye, testing=False)
    if amb_eye is None:
        amb_eye = np.zeros_like(amb_aud)
    comp = O.composite_rays_train_forward("triplane", sigmas, rgbs, deltas, rays, amb0=np.abs(amb_aud).sum(-1),
                                          amb1=np.abs(amb_eye).sum(-1), unc=unc.reshape(-1), T_thresh=T_thresh)
    image = comp["image"] + (F32(1) - comp["weights_sum"])[:, None] * F32(bg_color)
    image = np.clip(image, F32(0), F32(1)).astype(F32)
    d = np.clip(comp["depth"] - nears, F32(0), None) / (fars - nears)               # renderer.py:385
    return dict(image=image, depth=d.astype(F32), comp=comp, xyzs=xyzs, dirs=dirs, deltas=deltas, rays=rays,
                counter=counter, sigmas=sigmas, rgbs=rgbs, amb_aud=amb_aud, amb_eye=amb_eye, unc=unc)
