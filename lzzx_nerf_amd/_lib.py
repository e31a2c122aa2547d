"""ctypes binding of liblzzx_nerf_hip.so -- the only way product code reaches the kernels.

Fails loudly: a missing library or a missing symbol raises at import/first use; there is no CPU or
PyTorch fallback anywhere in this package (the CPU checker lives in /oracle and is never imported here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LZZX_NERF_HIP_SO: another build of the same library (an experiment variant of lzzx_nerf_amd/build.py --variant); same ABI check applies
SO_PATH = os.environ.get("LZZX_NERF_HIP_SO") or os.path.join(_HERE, "lib", "liblzzx_nerf_hip.so")

vp, u32, f32, i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int


class HeadParams(C.Structure):
    """mirror of lz_head_params (include/lzzx_nerf_hip.h)"""
    _fields_ = [("emb_xy", vp), ("emb_yz", vp), ("emb_xz", vp), ("offsets", vp), ("packed", vp), ("enc_a", vp),
                ("ind_code", vp), ("eye", vp), ("bound", f32), ("S", f32), ("H", u32), ("testing", i32), ("precision", i32)]


class HeadBwdOut(C.Structure):
    """mirror of lz_head_bwd_out (include/lzzx_nerf_hip.h)"""
    _fields_ = [("denc", vp), ("small", vp), ("rec", vp)]


class TorsoParams(C.Structure):
    """mirror of lz_torso_params (include/lzzx_nerf_hip.h)"""
    _fields_ = [("deform_w0", vp), ("deform_w1", vp), ("deform_w2", vp), ("torso_w0", vp), ("torso_w1", vp), ("torso_w2", vp),
                ("emb", vp), ("offsets", vp), ("enc_anchor", vp), ("ind_code", vp), ("ind_dim", u32), ("gridtype", u32),
                ("torso_shrink", f32), ("S", f32), ("H", u32), ("density_grid", vp), ("G", u32), ("density_thresh", f32)]


class AudioParams(C.Structure):
    """mirror of lz_audio_params (include/lzzx_nerf_hip.h)"""
    _fields_ = [("c_w", vp * 4), ("c_b", vp * 4), ("fc_w", vp * 2), ("fc_b", vp * 2), ("ac_w", vp * 5), ("ac_b", vp * 5),
                ("al_w", vp), ("al_b", vp), ("dim_in", u32), ("dim_aud", u32), ("n_win", u32), ("use_att", u32)]


class Frame(C.Structure):
    """mirror of lz_frame (include/lzzx_nerf_hip.h)"""
    _fields_ = [("head", HeadParams), ("state", vp), ("workspace", vp), ("rays_alive", vp * 2), ("rays_t", vp), ("rays_o", vp),
                ("rays_d", vp), ("nears", vp), ("fars", vp), ("grid", vp), ("xyzs", vp), ("dirs", vp), ("deltas", vp), ("sigmas", vp),
                ("rgbs", vp), ("amb_aud", vp), ("amb_eye", vp), ("unc", vp), ("weights_sum", vp), ("depth", vp), ("image", vp),
                ("amb_aud_sum", vp), ("amb_eye_sum", vp), ("unc_sum", vp), ("ray_counts", vp), ("N", u32), ("max_steps", u32),
                ("C", u32), ("H", u32), ("bound", f32), ("dt_gamma", f32), ("T_thresh", f32),
                ("sample_budget", u32), ("n_step_cap", u32)]


class FrameFused(C.Structure):
    """mirror of lz_frame_fused (include/lzzx_nerf_hip.h)"""
    _fields_ = [("head", HeadParams), ("rays_o", vp), ("rays_d", vp), ("grid", vp), ("aabb", vp), ("nears", vp), ("fars", vp), ("rays_t", vp),
                ("order", vp), ("state", vp), ("keys", vp), ("weights_sum", vp), ("depth", vp), ("image", vp), ("amb_aud_sum", vp),
                ("amb_eye_sum", vp), ("unc_sum", vp), ("out", vp), ("bg", vp), ("out_rgb24", vp), ("ray_counts", vp),
                ("bg_scalar", f32), ("bound", f32), ("dt_gamma", f32), ("T_thresh", f32), ("min_near", f32),
                ("N", u32), ("max_steps", u32), ("C", u32), ("H", u32), ("steps_per_pass", u32), ("noises", vp), ("occupied_aabb", vp), ("t_end", vp),
                ("cap_mode", u32), ("N_total", u32), ("defer_finish", u32), ("ray_last", vp), ("cap_ws", vp)]


class FrameNgp(C.Structure):
    """mirror of lz_frame_ngp (include/lzzx_nerf_hip.h)"""
    _fields_ = [("packed", vp), ("embeddings", vp), ("offsets", vp), ("enc_L", u32), ("enc_H", u32), ("enc_S", f32), ("emb_f16", i32), ("feats", vp),
                ("state", vp), ("workspace", vp), ("rays_alive", vp * 2), ("rays_t", vp), ("rays_o", vp), ("rays_d", vp), ("nears", vp), ("fars", vp),
                ("grid", vp), ("xyzs", vp), ("dirs", vp), ("deltas", vp), ("sigmas", vp), ("rgbs", vp), ("weights_sum", vp), ("depth", vp),
                ("image", vp), ("ray_counts", vp), ("N", u32), ("max_steps", u32), ("C", u32), ("H", u32), ("bound", f32), ("dt_gamma", f32),
                ("T_thresh", f32), ("sample_budget", u32), ("n_step_cap", u32)]


# name -> argtypes, in the order of include/lzzx_nerf_hip.h
SIGNATURES = {
    "lz_grid_encode_forward": [vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, vp, u32, i32, i32, i32, vp],
    "lz_grid_encode_backward": [vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, vp, vp, u32, i32, i32, i32, vp],
    "lz_grid_corner_indices": [vp, vp, vp, u32, u32, u32, u32, f32, u32, u32, i32, vp],
    "lz_sh_encode_forward": [vp, vp, u32, u32, u32, vp, vp],
    "lz_sh_encode_backward": [vp, vp, u32, u32, u32, vp, vp, vp],
    "lz_freq_encode_forward": [vp, u32, u32, u32, u32, vp, vp],
    "lz_freq_encode_backward": [vp, vp, u32, u32, u32, u32, vp, vp],
    "lz_near_far_from_aabb": [vp, vp, vp, u32, f32, vp, vp, vp],
    "lz_sph_from_ray": [vp, vp, f32, u32, vp, vp],
    "lz_morton3D": [vp, u32, vp, vp],
    "lz_morton3D_invert": [vp, u32, vp, vp],
    "lz_packbits": [vp, u32, f32, vp, vp],
    "lz_morton3D_dilation": [vp, u32, u32, vp, vp],
    "lz_march_rays_train": [vp, vp, vp, f32, f32, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "lz_march_rays_train_backward": [vp, vp, vp, vp, u32, u32, vp, vp, vp],
    "lz_march_rays": [u32, u32, vp, vp, vp, vp, f32, f32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp],
    "lz_composite_train_forward_v": [vp, vp, vp, vp, vp, vp, vp, u32, u32, f32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    "lz_composite_train_backward_v": [vp] * 16 + [u32, u32, f32, i32, i32, i32, i32] + [vp] * 6,
    # step-major sample rows for training (round 5): 64 neighbouring rays at the same step per wave of the consumers
    "lz_march_rays_train_grouped": [vp, vp, vp, f32, f32, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "lz_march_rays_train_backward_grouped": [vp, vp, vp, vp, u32, u32, vp, vp, vp],
    "lz_ray_sort_keys": [vp, vp, u32, f32, vp, vp],
    "lz_composite_rays_v": [u32, u32, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    # the reference's 13 compositing entry points by name (thin wrappers over the three *_v entries above)
    "lz_composite_rays_train_forward": [vp] * 5 + [u32, u32, f32] + [vp] * 5,
    "lz_composite_rays_train_backward": [vp] * 11 + [u32, u32, f32] + [vp] * 4,
    "lz_composite_rays": [u32, u32, f32] + [vp] * 9,
    "lz_composite_rays_ambient": [u32, u32, f32] + [vp] * 11,
    "lz_composite_rays_train_sigma_forward": [vp] * 5 + [u32, u32, f32] + [vp] * 5,
    "lz_composite_rays_train_sigma_backward": [vp] * 11 + [u32, u32, f32] + [vp] * 4,
    "lz_composite_rays_ambient_sigma": [u32, u32, f32] + [vp] * 11,
    "lz_composite_rays_train_uncertainty_forward": [vp] * 6 + [u32, u32, f32] + [vp] * 6,
    "lz_composite_rays_train_uncertainty_backward": [vp] * 14 + [u32, u32, f32] + [vp] * 5,
    "lz_composite_rays_uncertainty": [u32, u32, f32] + [vp] * 13,
    "lz_composite_rays_train_triplane_forward": [vp] * 7 + [u32, u32, f32] + [vp] * 7,
    "lz_composite_rays_train_triplane_backward": [vp] * 17 + [u32, u32, f32] + [vp] * 6,
    "lz_composite_rays_triplane": [u32, u32, f32] + [vp] * 15,
    "lz_get_rays": [vp, f32, f32, f32, f32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp],
    "lz_bg_coords": [u32, u32, vp, vp],
    "lz_head_pack_weights": [vp] * 11 + [i32, i32, vp, vp],
    "lz_triplane_head_forward": [C.POINTER(HeadParams), vp, vp, u32, vp, vp, vp, vp, vp, vp, vp],
    "lz_perturb_starts": [vp, vp, f32, u32, u32, u32, u32, vp, vp],
    "lz_wait_flags": [vp, u32, i32, u32, vp, vp],
    "lz_loop_begin": [u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "lz_loop_march": [vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, f32, f32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp],
    "lz_loop_composite": [vp, u32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "lz_loop_composite_plain": [vp, u32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "lz_loop_run": [C.POINTER(Frame), u32, u32, vp, vp],
    "lz_grid_encode_forward_tiled": [vp, vp, vp, vp, u32, vp, f32, u32, u32, u32, f32, u32, u32, i32, i32, vp],
    "lz_ngp_head_forward": [vp, vp, i32, vp, u32, vp, vp, vp, vp],
    "lz_ngp_loop_run": [C.POINTER(FrameNgp), u32, u32, vp],
    "lz_timing_create": [u32, C.POINTER(vp)],
    "lz_timing_destroy": [vp],
    "lz_timing_reset": [vp],
    "lz_timing_mark": [vp, i32, vp],
    "lz_timing_elapsed_ms": [vp, C.POINTER(f32), u32, C.POINTER(u32)],
    "lz_frame_render": [C.POINTER(FrameFused), vp, vp],
    "lz_frame_finish": [C.POINTER(FrameFused), vp],
    "lz_occupied_bounds": [vp, u32, u32, f32, u32, vp, vp, vp],
    "lz_final_blend": [vp, vp, vp, f32, u32, vp, vp],
    "lz_final_blend_rgb24": [vp, vp, vp, f32, u32, vp, vp, vp],
    "lz_debug_head_clocks": [C.POINTER(C.c_uint64)],
    "lz_triplane_head_backward": [C.POINTER(HeadParams), vp, vp, u32, vp, vp, vp, vp, vp, C.POINTER(HeadBwdOut), vp],
    "lz_head_pack_weights_f16": [vp] * 9 + [i32, i32, vp, vp],
    "lz_head_pack_weights_f16w": [vp] * 9 + [i32, i32, vp, vp],
    "lz_torso_forward": [C.POINTER(TorsoParams), vp, u32, vp, vp, vp, vp],
    "lz_torso_anchor_encode": [vp, vp, vp, vp],
    "lz_audio_encode": [C.POINTER(AudioParams), vp, vp, vp, vp],
    "lz_mark_untrained_grid": [vp, u32, f32, f32, f32, f32, u32, u32, f32, vp, vp, vp],
    "lz_density_grid_points": [vp, u32, u32, f32, vp, vp],
    "lz_density_grid_torso_points": [vp, u32, vp, vp],
    "lz_density_grid_torso_update": [vp, f32, f32, u32, vp, vp, vp, vp],
    "lz_density_grid_update": [vp, f32, f32, f32, u32, u32, vp, vp, vp, vp, vp],
    "lz_linear_forward": [vp, u32, vp, vp, u32, vp, u32, u32, u32, u32, i32, vp],
    "lz_linear_grad_w": [vp, u32, vp, vp, u32, vp, u32, u32, u32, u32, vp],
    "lz_triplane_head_forward_record": [C.POINTER(HeadParams), vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, i32, vp],
    "lz_triplane_head_backward_recorded": [C.POINTER(HeadParams), vp, u32, vp, vp, vp, vp, vp, C.POINTER(HeadBwdOut), i32, vp, vp],
    "lz_head_pack_weights_bwd_f16": [vp] * 7 + [i32, i32, vp, vp],
    "lz_triplane_plane_coords": [vp, u32, f32, vp, vp],
    "lz_head_pack_unc_f16": [vp, vp, vp, vp],
    "lz_triplane_head_forward_record_f16": [C.POINTER(HeadParams), vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp],
    "lz_triplane_head_forward_encx_f16": [C.POINTER(HeadParams), vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp],
    "lz_triplane_head_backward_encx_dw16": [C.POINTER(HeadParams), vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, C.POINTER(HeadBwdOut), vp, u32, vp, vp, vp, vp, vp,
                                            vp, vp],
    "lz_triplane_head_grad_w_f16": [vp, u32, u32, vp, vp, vp, vp, vp, vp, vp],
    "lz_triplane_head_backward_recorded_dw16": [C.POINTER(HeadParams), vp, vp, u32, vp, vp, vp, vp, vp, C.POINTER(HeadBwdOut), vp, u32, vp, vp, vp, vp, vp,
                                                vp, vp],
    "lz_triplane_head_grad_w": [vp, u32, u32, vp, vp, vp, vp, vp, vp, vp],
}
PLAIN = {"lz_last_error": ([], C.c_char_p), "lz_abi_version": ([], i32), "lz_device_ok": ([], i32), "lz_train_group_size": ([], i32),
         "lz_head_packed_size": ([], u32), "lz_head_packed_size_f16": ([], u32), "lz_head_packed_size_f16w": ([], u32), "lz_head_packed_unc_size_f16": ([], u32), "lz_head_packed_bwd_size_f16": ([], u32),
         "lz_triplane_head_grad_w_workspace": ([], C.c_size_t)}

ALL_SYMBOLS = sorted(list(SIGNATURES) + list(PLAIN))
ABI_VERSION = 11  # lz_abi_version() of the library this binding table describes (include/lzzx_nerf_hip.h)

_lib = None


class LzError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every symbol of the header; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise LzError(
            "liblzzx_nerf_hip.so not found at %s -- build it with `python -m lzzx_nerf_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    older = bool(os.environ.get("LZZX_NERF_HIP_SO_OLDER"))   # A/B runs against a library of an EARLIER round (tools/ab_rounds.sh): entry points it lacks stay unbound
    for name, argtypes in SIGNATURES.items():
        if older and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = i32
    for name, (argtypes, restype) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().lz_last_error()
        raise LzError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else ""))


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
