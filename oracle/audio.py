"""TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  CPU restatement of NeRFNetwork.encode_audio
(/root/reference/nerf_triplane/network.py:226-240): AudioNet (network.py:40-70) + AudioAttNet (network.py:9-37), float32 fma
chains in the order csrc/lz_audio.hip uses (input channel outer, tap inner, bias after the chain)."""
import numpy as np

from . import oracle as O

F32 = np.float32


def _lrelu(v):
    return np.where(v > 0, v, F32(0.02) * v).astype(F32)


def conv1d_k3(x, w, b, stride):
    """x [n, Cin, Lin], w [Cout, Cin, 3], padding 1 -> lrelu(conv + b) [n, Cout, Lout]"""
    n, Cin, Lin = x.shape
    Cout = w.shape[0]
    Lout = (Lin - 1) // stride + 1
    acc = np.zeros((n, Cout, Lout), F32)
    t = np.arange(Lout)
    for ci in range(Cin):
        for k in range(3):
            pos = t * stride + k - 1
            ok = (pos >= 0) & (pos < Lin)
            xv = np.where(ok[None, :], x[:, ci, np.clip(pos, 0, Lin - 1)], F32(0)).astype(F32)          # [n, Lout]
            new = O.fma(np.broadcast_to(w[None, :, ci, k, None], acc.shape), np.broadcast_to(xv[:, None, :], acc.shape), acc)
            acc = np.where(ok[None, None, :], new, acc)                                                  # skipped taps leave acc untouched
    return _lrelu(acc + b[None, :, None].astype(F32))


def conv1_wide(x, w, b):
    """first AudioNet layer in the order csrc/lz_audio.hip uses for dim_in >= 128 (lz_k_audio_conv1_wide): lane l of a wave owns
    channels l, l + 64, ... (fma chain, channel outer, tap inner); the 64 partials are combined by an xor-shuffle tree"""
    n, Cin, Lin = x.shape
    Cout, Lout = w.shape[0], 8
    part = np.zeros((64, n, Cout, Lout), F32)
    t = np.arange(Lout)
    for ci in range(Cin):
        l = ci % 64
        for k in range(3):
            pos = t * 2 + k - 1
            ok = (pos >= 0) & (pos < Lin)
            xv = np.where(ok[None, :], x[:, ci, np.clip(pos, 0, Lin - 1)], F32(0)).astype(F32)
            new = O.fma(np.broadcast_to(w[None, :, ci, k, None], part[l].shape), np.broadcast_to(xv[:, None, :], part[l].shape), part[l])
            part[l] = np.where(ok[None, None, :], new, part[l])
    off = 32
    while off:   # acc += shfl_xor(acc, off): every lane adds its partner; lane 0 ends with the tree sum
        part = (part + part[np.arange(64) ^ off]).astype(F32)
        off >>= 1
    return _lrelu(part[0] + b[None, :, None].astype(F32))


def fc(x, w, b, lrelu):
    y = O.linear(np.ascontiguousarray(x, F32), np.ascontiguousarray(w, F32)) + b[None, :].astype(F32)
    return _lrelu(y) if lrelu else y.astype(F32)


def encode_audio(P, a, use_att=True):
    """a [n_win, dim_in, 16] -> enc_a [1, dim_aud] (attention) or [n_win, dim_aud]"""
    x = np.ascontiguousarray(a, F32)
    for i, s in zip((0, 2, 4, 6), (2, 2, 2, 2)):
        if i == 0 and x.shape[1] >= 128:
            x = conv1_wide(x, P["audio_net.encoder_conv.0.weight"], P["audio_net.encoder_conv.0.bias"])
            continue
        x = conv1d_k3(x, P["audio_net.encoder_conv.%d.weight" % i], P["audio_net.encoder_conv.%d.bias" % i], s)
    x = x[:, :, 0]
    x = fc(x, P["audio_net.encoder_fc1.0.weight"], P["audio_net.encoder_fc1.0.bias"], True)
    feat = fc(x, P["audio_net.encoder_fc1.2.weight"], P["audio_net.encoder_fc1.2.bias"], False)       # [n, dim_aud]
    if not use_att:
        return feat
    y = np.ascontiguousarray(feat.T[None])                                                             # [1, dim_aud, n]
    for i in (0, 2, 4, 6, 8):
        y = conv1d_k3(y, P["audio_att_net.attentionConvNet.%d.weight" % i], P["audio_att_net.attentionConvNet.%d.bias" % i], 1)
    logits = fc(y.reshape(1, -1), P["audio_att_net.attentionNet.0.weight"], P["audio_att_net.attentionNet.0.bias"], False)[0]
    m = logits.max()
    e = O.unary("exp", (logits - m).astype(F32))
    s = F32(0)
    for v in e:
        s = F32(s + v)
    wgt = (e / s).astype(F32)
    acc = np.zeros(feat.shape[1], F32)
    for t in range(feat.shape[0]):
        acc = O.fma(np.full_like(acc, wgt[t]), feat[t], acc)
    return acc[None]
