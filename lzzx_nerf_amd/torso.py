"""Torso branch of a frame as one kernel (csrc/lz_torso.hip): `NeRFRenderer.run_torso`'s masked query
(/root/reference/nerf_triplane/renderer.py:572-631) + `NeRFNetwork.forward_torso` (/root/reference/nerf_triplane/network.py:170-205).

It consumes the reference's state_dict unchanged: `anchor_points`, `torso_deform_net.net.{0,1,2}.weight`,
`torso_encoder.{embeddings,offsets}`, `torso_net.net.{0,1,2}.weight`, `individual_codes_torso`.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._util import call, ptr, stream
from .freqencoder import FreqEncoder


class FusedTorso:
    def __init__(self, state_dict, torso_shrink=0.8, device="cuda"):
        self.device = torch.device(device)
        g = lambda k: state_dict[k].detach().to(self.device, torch.float32).contiguous()
        self.anchor_points = g("anchor_points")                                     # [3, 4]
        self.dw = [g("torso_deform_net.net.%d.weight" % i) for i in range(3)]
        self.tw = [g("torso_net.net.%d.weight" % i) for i in range(3)]
        self.emb = g("torso_encoder.embeddings")
        self.offsets = state_dict["torso_encoder.offsets"].to(self.device, torch.int32).contiguous()
        self.ind_dim = self.dw[0].shape[1] - 34 - 42
        if self.ind_dim not in (0, 8) or self.tw[0].shape[1] != 32 + 34 + 42 + self.ind_dim or self.offsets.numel() != 17 or self.emb.shape[1] != 2:
            raise RuntimeError("FusedTorso expects the reference's torso configuration (network.py:156-167)")
        self.torso_shrink = float(torso_shrink)
        # GridEncoder hyper-parameters of the torso encoder (network.py:166, grid.py:95-96): H = 16, desired 2048, L = 16
        self.H = 16
        self.S = float(np.float32(np.log2(np.exp2(np.log2(2048 / 16) / 15))))
        self.anchor_encoder = FreqEncoder(input_dim=6, degree=3)

    def encode_anchor(self, poses):
        """network.py:179-183: anchor points warped by the inverse head pose, perspective-divided, frequency-encoded -> [1, 42]"""
        pose = poses.to(self.device, torch.float32).reshape(-1, 4, 4)
        if pose.shape[0] != 1:
            raise RuntimeError("encode_anchor: one head pose per frame (the reference's forward_torso asserts the same, network.py:179)")
        pose = pose.contiguous()
        enc = torch.empty(1, 42, dtype=torch.float32, device=self.device)
        call("lz_torso_anchor_encode", ptr(pose), ptr(self.anchor_points), ptr(enc), stream())   # one launch; torch.inverse alone is a dozen
        return enc

    @torch.no_grad()
    def forward(self, bg_coords, poses=None, ind_code=None, density_grid=None, density_thresh=0.0, enc_anchor=None):
        """bg_coords [N,2] in [-1,1].  Returns torso_alpha [N,1], torso_color [N,3], deform [N,2]; pixels whose 2-D occupancy
        (bilinear sample of density_grid [G*G]) is <= density_thresh get zeros, as in run_torso."""
        xy = bg_coords.reshape(-1, 2).to(self.device, torch.float32).contiguous()
        N = xy.shape[0]
        if enc_anchor is None:
            enc_anchor = self.encode_anchor(poses)
        enc_anchor = enc_anchor.reshape(-1).float().contiguous()
        if self.ind_dim:
            if ind_code is None:
                raise RuntimeError("this torso network was trained with an individual code (ind_dim_torso = %d)" % self.ind_dim)
            ind_code = ind_code.reshape(-1).to(self.device, torch.float32).contiguous()
        p = _lib.TorsoParams()
        p.deform_w0, p.deform_w1, p.deform_w2 = [w.data_ptr() for w in self.dw]
        p.torso_w0, p.torso_w1, p.torso_w2 = [w.data_ptr() for w in self.tw]
        p.emb, p.offsets, p.enc_anchor = self.emb.data_ptr(), self.offsets.data_ptr(), enc_anchor.data_ptr()
        p.ind_code = ind_code.data_ptr() if self.ind_dim else None
        p.ind_dim, p.gridtype, p.torso_shrink, p.S, p.H = self.ind_dim, 1, self.torso_shrink, self.S, self.H
        if density_grid is not None:
            density_grid = density_grid.reshape(-1).to(self.device, torch.float32).contiguous()
            G = round(density_grid.numel() ** 0.5)
            p.density_grid, p.G, p.density_thresh = density_grid.data_ptr(), G, float(density_thresh)
        else:
            p.density_grid, p.G, p.density_thresh = None, 0, 0.0
        kw = dict(dtype=torch.float32, device=self.device)
        alpha, color, deform = torch.empty(N, 1, **kw), torch.empty(N, 3, **kw), torch.empty(N, 2, **kw)
        call("lz_torso_forward", C.byref(p), ptr(xy), N, ptr(alpha), ptr(color), ptr(deform), stream())
        self._keep = (enc_anchor, ind_code, density_grid)
        return alpha, color, deform

    __call__ = forward

    @staticmethod
    def mix_background(alpha, color, bg_color):
        """renderer.py:621: bg = torso_color * torso_alpha + bg_color * (1 - torso_alpha)"""
        return color * alpha + bg_color * (1 - alpha)
