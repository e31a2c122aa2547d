"""Audio front-end: FusedAudioEncoder (one launch) vs the CPU restatement of encode_audio, bit for bit."""
import numpy as np
import pytest
import torch

from test_audio_oracle import audio_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim_in,att", [(29, True), (44, True), (1024, True), (29, False)])
def test_fused_audio_encoder_matches_checker(dim_in, att):
    from lzzx_nerf_amd.audio import FusedAudioEncoder
    from oracle.audio import encode_audio
    sd = audio_state(dim_in, 32, att)
    rng = np.random.default_rng(dim_in)
    n = 8 if att else 1
    a = rng.normal(size=(n, dim_in, 16)).astype(np.float32)
    enc = FusedAudioEncoder({k: torch.from_numpy(v) for k, v in sd.items()})
    out = enc(torch.from_numpy(a).cuda()).cpu().numpy()
    ref = encode_audio(sd, a, att)
    assert out.shape == ref.shape == ((1, 32) if att else (n, 32))
    assert np.array_equal(out, ref)
    with pytest.raises(RuntimeError, match="must be"):
        enc(torch.zeros(n, dim_in + 1, 16))
