"""Torso branch for TRAINING: `NeRFNetwork.forward_torso` (/root/reference/nerf_triplane/network.py:170-205) as an autograd graph over
this repo's operators -- frequency encoder (csrc/lz_encoders.hip, forward + backward), tiled-grid encoder D = 2 / L = 16 / C = 2
(csrc/lz_grid.hip, forward + scatter-add backward + dy/dx for the deformation path) and the bias-free MLPs on the MFMA Linear kernels
(csrc/lz_linear.hip).  Same parameters and state-dict keys as the reference (`anchor_points`, `torso_deform_net.*`, `torso_encoder.*`,
`torso_net.*`), so a checkpoint moves between this module, the reference and the one-launch inference kernel (`torso.FusedTorso`)
unchanged.  The audio nets (`AudioNet`, `AudioAttNet`) are plain torch Conv1d / Linear modules in the reference and train as they are;
only their fused INFERENCE kernel (`audio.FusedAudioEncoder`) is forward-only."""
import torch
import torch.nn as nn

from .encoding import get_encoder
from .linear import MLP


class TorsoTrainNet(nn.Module):
    def __init__(self, ind_dim_torso=8, torso_shrink=0.8):
        super().__init__()
        self.torso_shrink = float(torso_shrink)
        self.anchor_points = nn.Parameter(torch.tensor([[0.01, 0.01, 0.1, 1], [-0.1, -0.1, 0.1, 1], [0.1, -0.1, 0.1, 1]]))   # network.py:158-159
        self.torso_deform_encoder, d_in = get_encoder("frequency", input_dim=2, multires=8)
        self.anchor_encoder, a_in = get_encoder("frequency", input_dim=6, multires=3)
        self.torso_deform_net = MLP(d_in + a_in + ind_dim_torso, 2, 32, 3)
        self.torso_encoder, t_in = get_encoder("tiledgrid", input_dim=2, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                                               desired_resolution=2048)
        self.torso_net = MLP(t_in + d_in + a_in + ind_dim_torso, 4, 32, 3)

    def forward(self, x, poses, c=None):
        """x [N,2] in [-1,1]; poses [1,4,4]; c [1, ind_dim_torso] or None -> alpha [N,1], color [N,3], dx [N,2]"""
        x = x * self.torso_shrink
        wrapped = self.anchor_points[None, ...] @ poses.permute(0, 2, 1).inverse()
        wrapped = (wrapped[:, :, :2] / wrapped[:, :, 3, None] / wrapped[:, :, 2, None]).view(1, -1)
        enc_anchor = self.anchor_encoder(wrapped)
        enc_x = self.torso_deform_encoder(x)
        parts = [enc_x, enc_anchor.repeat(x.shape[0], 1)]
        if c is not None:
            parts.append(c.repeat(x.shape[0], 1))
        h = torch.cat(parts, dim=-1)
        dx = self.torso_deform_net(h)
        x = (x + dx).clamp(-1, 1)
        x = self.torso_encoder(x, bound=1)
        h = self.torso_net(torch.cat([x, h], dim=-1))
        alpha = torch.sigmoid(h[..., :1]) * (1 + 2 * 0.001) - 0.001
        color = torch.sigmoid(h[..., 1:]) * (1 + 2 * 0.001) - 0.001
        return alpha, color, dx
