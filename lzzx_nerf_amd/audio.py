"""Audio conditioning front-end on device: `NeRFNetwork.encode_audio` (/root/reference/nerf_triplane/network.py:226-240), i.e.
`AudioNet` (network.py:40-70) + `AudioAttNet` (network.py:9-37), as ONE kernel launch (csrc/lz_audio.hip).

Consumes the reference's state_dict keys unchanged: `audio_net.encoder_conv.{0,2,4,6}.{weight,bias}`,
`audio_net.encoder_fc1.{0,2}.{weight,bias}`, `audio_att_net.attentionConvNet.{0,2,4,6,8}.{weight,bias}`,
`audio_att_net.attentionNet.0.{weight,bias}`.
"""
import ctypes as C

import torch

from . import _lib
from ._util import call, ptr, stream


class FusedAudioEncoder:
    def __init__(self, state_dict, device="cuda"):
        self.device = torch.device(device)
        g = lambda k: state_dict[k].detach().to(self.device, torch.float32).contiguous()
        self.cw = [g("audio_net.encoder_conv.%d.weight" % i) for i in (0, 2, 4, 6)]
        self.cb = [g("audio_net.encoder_conv.%d.bias" % i) for i in (0, 2, 4, 6)]
        self.fw = [g("audio_net.encoder_fc1.%d.weight" % i) for i in (0, 2)]
        self.fb = [g("audio_net.encoder_fc1.%d.bias" % i) for i in (0, 2)]
        self.use_att = "audio_att_net.attentionNet.0.weight" in state_dict      # opt.att > 0 (network.py:124-126)
        if self.use_att:
            self.aw = [g("audio_att_net.attentionConvNet.%d.weight" % i) for i in (0, 2, 4, 6, 8)]
            self.ab = [g("audio_att_net.attentionConvNet.%d.bias" % i) for i in (0, 2, 4, 6, 8)]
            self.lw, self.lb = g("audio_att_net.attentionNet.0.weight"), g("audio_att_net.attentionNet.0.bias")
        self.dim_in, self.dim_aud = self.cw[0].shape[1], self.fw[1].shape[0]
        if tuple(self.cw[0].shape) != (32, self.dim_in, 3) or tuple(self.cw[3].shape) != (64, 64, 3) or self.dim_aud > 64:
            raise RuntimeError("FusedAudioEncoder expects the reference's AudioNet layout (network.py:45-60)")

    @torch.no_grad()
    def forward(self, a):
        """a: [n_win, dim_in, 16] audio feature windows (n_win = 8 with attention, 1 without).  Returns enc_a [1, dim_aud]
        (attention) or [n_win, dim_aud]."""
        a = a.to(self.device, torch.float32).contiguous()
        n = a.shape[0]
        if a.dim() != 3 or a.shape[1] != self.dim_in or a.shape[2] != 16:
            raise RuntimeError("audio features must be [n_win, %d, 16]" % self.dim_in)
        p = _lib.AudioParams()
        for i in range(4):
            p.c_w[i], p.c_b[i] = self.cw[i].data_ptr(), self.cb[i].data_ptr()
        for i in range(2):
            p.fc_w[i], p.fc_b[i] = self.fw[i].data_ptr(), self.fb[i].data_ptr()
        if self.use_att:
            if self.lw.shape[0] != n:
                raise RuntimeError("AudioAttNet was built for %d windows, got %d" % (self.lw.shape[0], n))
            for i in range(5):
                p.ac_w[i], p.ac_b[i] = self.aw[i].data_ptr(), self.ab[i].data_ptr()
            p.al_w, p.al_b = self.lw.data_ptr(), self.lb.data_ptr()
        p.dim_in, p.dim_aud, p.n_win, p.use_att = self.dim_in, self.dim_aud, n, int(self.use_att)
        out = torch.empty((1 if self.use_att else n), self.dim_aud, dtype=torch.float32, device=self.device)
        ws = torch.empty(n * 256, dtype=torch.float32, device=self.device) if self.dim_in >= 128 else None
        call("lz_audio_encode", C.byref(p), ptr(a), ptr(out), ptr(ws), stream())
        return out

    __call__ = forward
