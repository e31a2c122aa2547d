"""CPU: oracle/audio.py (restatement of AudioNet + AudioAttNet, network.py:9-70, 226-240) pinned to torch's own Conv1d / Linear /
LeakyReLU / Softmax modules arranged as the reference arranges them (tolerance = summation order)."""
import numpy as np
import torch
import torch.nn as nn


def audio_state(dim_in, dim_aud, att, seed=0):
    """state-dict-shaped random weights with the reference's key names and shapes"""
    g = torch.Generator().manual_seed(seed + dim_in)
    sd = {}
    def conv(prefix, idx, cin, cout):
        k = 1.0 / np.sqrt(cin * 3)
        sd[f"{prefix}.{idx}.weight"] = ((torch.rand(cout, cin, 3, generator=g) * 2 - 1) * k).numpy()
        sd[f"{prefix}.{idx}.bias"] = ((torch.rand(cout, generator=g) * 2 - 1) * k).numpy()
    def lin(prefix, idx, cin, cout):
        k = 1.0 / np.sqrt(cin)
        sd[f"{prefix}.{idx}.weight"] = ((torch.rand(cout, cin, generator=g) * 2 - 1) * k).numpy()
        sd[f"{prefix}.{idx}.bias"] = ((torch.rand(cout, generator=g) * 2 - 1) * k).numpy()
    for idx, (ci, co) in zip((0, 2, 4, 6), ((dim_in, 32), (32, 32), (32, 64), (64, 64))):
        conv("audio_net.encoder_conv", idx, ci, co)
    lin("audio_net.encoder_fc1", 0, 64, 64)
    lin("audio_net.encoder_fc1", 2, 64, dim_aud)
    if att:
        for idx, (ci, co) in zip((0, 2, 4, 6, 8), ((dim_aud, 16), (16, 8), (8, 4), (4, 2), (2, 1))):
            conv("audio_att_net.attentionConvNet", idx, ci, co)
        lin("audio_att_net.attentionNet", 0, 8, 8)
    return sd


def _torch_encode_audio(sd, a, att):
    t = lambda k: torch.from_numpy(sd[k]).double()
    x = torch.from_numpy(a).double()
    for idx in (0, 2, 4, 6):
        x = nn.functional.leaky_relu(nn.functional.conv1d(x, t(f"audio_net.encoder_conv.{idx}.weight"), t(f"audio_net.encoder_conv.{idx}.bias"),
                                                          stride=2, padding=1), 0.02)
    x = x.squeeze(-1)
    x = nn.functional.leaky_relu(nn.functional.linear(x, t("audio_net.encoder_fc1.0.weight"), t("audio_net.encoder_fc1.0.bias")), 0.02)
    x = nn.functional.linear(x, t("audio_net.encoder_fc1.2.weight"), t("audio_net.encoder_fc1.2.bias"))
    if not att:
        return x.numpy()
    x = x.unsqueeze(0)                                    # [1, seq_len, dim_aud]
    y = x.permute(0, 2, 1)
    for idx in (0, 2, 4, 6, 8):
        y = nn.functional.leaky_relu(nn.functional.conv1d(y, t(f"audio_att_net.attentionConvNet.{idx}.weight"),
                                                          t(f"audio_att_net.attentionConvNet.{idx}.bias"), stride=1, padding=1), 0.02)
    y = torch.softmax(nn.functional.linear(y.view(1, 8), t("audio_att_net.attentionNet.0.weight"), t("audio_att_net.attentionNet.0.bias")), dim=1)
    return torch.sum(y.view(1, 8, 1) * x, dim=1).numpy()


def test_audio_oracle_matches_torch_modules():
    from oracle.audio import encode_audio
    for dim_in, att in ((29, True), (44, False), (128, True)):
        sd = audio_state(dim_in, 32, att)
        rng = np.random.default_rng(1)
        a = rng.normal(size=(8 if att else 3, dim_in, 16)).astype(np.float32)
        ref = _torch_encode_audio(sd, a, att)
        out = encode_audio(sd, a, att)
        assert out.shape == ref.shape and np.abs(out - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())
