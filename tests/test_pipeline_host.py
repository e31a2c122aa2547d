"""Host-side pieces of the frame pipeline that need no GPU: the audio window rule of the reference's data path."""
import numpy as np
import pytest
import torch

from lzzx_nerf_amd.pipeline import audio_window


def _window_restated(features, att_mode, index):
    """nerf_triplane/utils.py:20-52 restated with explicit loops"""
    n = features.shape[0]
    if att_mode == 0:
        return features[index:index + 1]
    lo, hi = (index - 8, index) if att_mode == 1 else (index - 4, index + 4)
    rows = []
    for i in range(lo, hi):
        rows.append(features[i] if 0 <= i < n else np.zeros_like(features[0]))
    return np.stack(rows)


@pytest.mark.parametrize("att_mode", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 3, 20])
def test_audio_window_matches_restated_rule(att_mode, n):
    feats = np.arange(n * 2 * 3, dtype=np.float32).reshape(n, 2, 3) + 1
    for index in range(n):
        got = audio_window(torch.from_numpy(feats), att_mode, index).numpy()
        want = _window_restated(feats, att_mode, index)
        assert got.shape == want.shape and np.array_equal(got, want), (att_mode, n, index)
    if att_mode:
        assert audio_window(torch.from_numpy(feats), att_mode, 0).shape[0] == 8


def test_audio_window_rejects_unknown_mode():
    with pytest.raises(NotImplementedError):
        audio_window(torch.zeros(4, 2), 3, 0)
