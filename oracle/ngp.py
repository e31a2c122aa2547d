"""CPU checker for the fused hash-grid NeRF head (lzzx_nerf_amd/ngp.py, csrc/lz_ngp.hip) -- TEST INFRASTRUCTURE ONLY.

The network is the operator-API one (get_encoder('hashgrid') defaults -> MLP 32-64-16 -> [SH(4) | 15 geometry features] -> MLP 31-64-3,
bias-free Linear + ReLU as /root/reference/nerf_triplane/network.py:73-94, encoders as encoding.py:6-37); what this file pins is the
summation ORDER of every Linear, which a torch `x @ W.T` leaves to the library and the kernel fixes by its MFMA operand layout:
  sigma_net.0   k-step ks = 2 i + c, lane q: feature 2 (q + 4 i) + c  (lane q holds levels q, q + 4, q + 8, q + 12, both channels)
  hidden layers "chained": the previous accumulator tile in place (oracle.head.korder_chained)
  colour_net.0  SH components in natural order, then sigma_net outputs 4 q + r for r, q (output 0 = sigma's row: not an input)"""
import numpy as np

from . import oracle as O
from .head import korder_chained, korder_natural

F32 = np.float32


def korder_levels():
    return [2 * (q + 4 * (ks >> 1)) + (ks & 1) for ks in range(8) for q in range(4)]


def korder_color0():
    k = korder_natural(16)
    for r in range(4):
        for q in range(4):
            slot = 4 * q + r
            k.append(16 + slot - 1 if slot >= 1 else -1)
    return k


def head_forward(W, feats, dirs):
    """W: dict s0 [64,32], s1 [16,64], c0 [64,31], c1 [3,64]; feats [M,32] f32 (the encoder's output, widened if the table is f16); dirs [M,3]"""
    h1 = O.linear(np.ascontiguousarray(feats, dtype=F32), W["s0"], korder_levels(), relu=True)
    h = O.linear(h1, W["s1"], korder_chained(64))
    sigma = O.unary("exp", np.ascontiguousarray(h[:, 0]))
    sh, _ = O.sh_encode_forward(np.ascontiguousarray(dirs, dtype=F32), 4)
    x = np.ascontiguousarray(np.concatenate([sh, h[:, 1:]], 1))
    c1 = O.linear(x, W["c0"], korder_color0(), relu=True)
    rgb = O.unary("sigmoid", O.linear(c1, W["c1"], korder_chained(64)))
    return sigma, rgb


def network(W, emb, offsets, per_level_scale, base_resolution=16):
    """(xyzs, dirs, bound) -> (sigma, rgb): GridEncoder.forward (grid.py:139-154) + head_forward; a half table yields half features"""
    def net(xyzs, dirs, bound):
        x01 = (np.asarray(xyzs, F32) + F32(bound)) / F32(2 * bound)
        feats, _ = O.grid_encode_forward(x01, emb, offsets, per_level_scale, base_resolution)
        return head_forward(W, feats.astype(F32), dirs)
    return net
