"""One inference frame end to end on device, in the order of `NeRFRenderer.run_cuda` / `run_cuda_for_inference`
(/root/reference/nerf_triplane/renderer.py:406-570):

    audio windows --encode_audio--> enc_a        (1 launch,  lzzx_nerf_amd.audio)
    bg_coords, head pose --torso--> background   (1 launch,  lzzx_nerf_amd.torso; skipped without torso weights)
    rays --march / head / composite loop--> rgb  (3 + 4 per iteration launches, lzzx_nerf_amd.renderer), blended over that background

All stages read the reference's state_dict; nothing synchronises with the host except the render loop's bounded look-ahead."""
import torch

from .audio import FusedAudioEncoder
from .head import FusedTriplaneHead
from .renderer import TriplaneRenderer
from .torso import FusedTorso


def audio_window(features, att_mode, index):
    """The per-frame slice of the audio feature track that feeds `encode_audio` (nerf_triplane/utils.py:20-52): att_mode 0 -> frame
    `index` alone, 1 -> the 8 frames before it, 2 -> frames [index - 4, index + 4); windows running off either end of the track
    are zero-padded to their full length (the reference pads with `zeros_like` of the rows it has, so for a track shorter than
    the window it returns fewer than 8 rows; tracks of at least 8 frames give identical windows)."""
    n = features.shape[0]
    if att_mode == 0:
        return features[[index]]
    if att_mode == 1:
        lo, hi = index - 8, index
    elif att_mode == 2:
        lo, hi = index - 4, index + 4
    else:
        raise NotImplementedError(f"wrong att_mode: {att_mode}")
    out = features.new_zeros((hi - lo,) + tuple(features.shape[1:]))
    a, b = max(lo, 0), min(hi, n)
    if b > a:
        out[a - lo: b - lo] = features[a:b]
    return out


class TalkingHeadFrame:
    SMOOTH_LIPS_LAMBDA = 0.35   # renderer.py:254-258, 456-460

    def __init__(self, state_dict, density_bitfield, bound=1.0, exp_eye=True, torso_shrink=0.8, precision="f32", device="cuda",
                 smooth_lips=False, fold_geo=False, **renderer_kw):
        self.smooth_lips = bool(smooth_lips)   # opt.smooth_lips: enc_a of a frame is blended with the previous frame's
        self._enc_a_prev = None
        self.audio = FusedAudioEncoder(state_dict, device=device)
        self.head = FusedTriplaneHead(state_dict, bound=bound, exp_eye=exp_eye, device=device, precision=precision, fold_geo=fold_geo)
        self.torso = FusedTorso(state_dict, torso_shrink=torso_shrink, device=device) if "torso_net.net.0.weight" in state_dict else None
        self.renderer = TriplaneRenderer(self.head, density_bitfield, bound=bound, **renderer_kw)

    @classmethod
    def from_checkpoint(cls, path_or_dict, density_thresh=10.0, density_thresh_torso=0.01, bitfield="auto", device="cuda", load_kwargs=None, **kw):
        """Build the frame pipeline from a checkpoint the reference's trainer wrote (TrainerUtil.py:1222-1281) or from a bare state dict
        (:1297-1300) -- see lzzx_nerf_amd/checkpoint.py for the two layouts.  `bound` and `exp_eye` default to what the tensors say;
        the occupancy bitfield is the stored buffer, else packbits(density_grid, min(mean_density, density_thresh)) (renderer.py:765),
        else all ones (`bitfield` = "auto" | "bitfield" | "grid" | "ones").  The running means travel with the object: `mean_count`
        (march_rays_train's buffer size, renderer.py:287), `mean_density`, `mean_density_torso`; with a torso grid in the file `render`
        masks the torso with `min(density_thresh_torso, mean_density_torso)` like run_torso (renderer.py:603) unless told otherwise."""
        from .checkpoint import infer_hyper, read_checkpoint, resolve_bitfield
        ck = read_checkpoint(path_or_dict, **(load_kwargs or {}))          # load_kwargs: map_location / weights_only for torch.load
        hyper = infer_hyper(ck.model)
        kw.setdefault("bound", hyper.get("bound", 1.0))
        kw.setdefault("exp_eye", hyper.get("exp_eye", True))
        if "aabb_infer" in ck.model:                                    # the box inference marches in (renderer.py:113, 476)
            kw.setdefault("aabb", ck.model["aabb_infer"])
        bits, grid, plan = resolve_bitfield(ck, density_thresh, device=device, source=bitfield)
        self = cls(ck.model, bits, device=device, **kw)
        self.checkpoint_kind, self.bitfield_plan = ck.kind, plan
        self.mean_count, self.mean_density, self.mean_density_torso = ck.mean_count, ck.mean_density, ck.mean_density_torso
        self.epoch, self.global_step = ck.epoch, ck.global_step
        self.density_grid = grid                                        # [cascade, G^3] on device (None for 'best' checkpoints)
        gt = ck.model.get("density_grid_torso")
        self.density_grid_torso = None if gt is None else gt.to(device=device, dtype=torch.float32).contiguous()
        self.density_thresh_torso = min(float(density_thresh_torso), ck.mean_density_torso)
        self.individual_codes = ck.model.get("individual_codes")
        return self

    def reset(self):
        """forget the previous frame's audio code (start of a new clip)"""
        self._enc_a_prev = None

    @torch.no_grad()
    def render(self, rays_o, rays_d, auds, eye=None, ind_code=None, bg_coords=None, poses=None, ind_code_torso=None, bg_color=1.0,
               density_grid_torso=None, density_thresh_torso=None, **render_kw):
        """auds [8, dim_in, 16]; bg_color: scalar or [N,3]; returns the renderer's dict plus enc_a and (with a torso) torso_alpha,
        torso_color (= the mixed background, as results['torso_color'] in run_torso) and deform."""
        enc_a = self.audio(auds)                                                            # renderer.py:455
        if self.smooth_lips:                                                                # renderer.py:456-460 (stateful across frames)
            if self._enc_a_prev is not None:
                enc_a = self.SMOOTH_LIPS_LAMBDA * self._enc_a_prev + (1 - self.SMOOTH_LIPS_LAMBDA) * enc_a
            self._enc_a_prev = enc_a
        extra = dict(enc_a=enc_a)
        if self.torso is not None and bg_coords is not None:
            if density_grid_torso is None:                      # from_checkpoint: the file's torso occupancy and its threshold
                density_grid_torso = getattr(self, "density_grid_torso", None)
            if density_thresh_torso is None:
                density_thresh_torso = getattr(self, "density_thresh_torso", 0.0) if density_grid_torso is not None else 0.0
            alpha, color, deform = self.torso(bg_coords, poses, ind_code_torso, density_grid=density_grid_torso,
                                              density_thresh=density_thresh_torso)         # renderer.py:572-617
            if not torch.is_tensor(bg_color):
                bg_color = torch.full_like(color, float(bg_color))
            bg_color = FusedTorso.mix_background(alpha, color, bg_color.reshape(-1, 3))     # renderer.py:621
            extra.update(torso_alpha=alpha, torso_color=bg_color, deform=deform)
        out = self.renderer.render(rays_o, rays_d, enc_a, ind_code, eye, bg_color=bg_color, **render_kw)
        out.update(extra)
        return out
