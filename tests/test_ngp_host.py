"""Host-side checks of the fused hash-grid NeRF path (lzzx_nerf_amd/ngp.py) that need no GPU: the packed MFMA fragments consume the
input features in exactly the summation order the checker spells (oracle/ngp.py), every weight lands in exactly one fragment slot, and
the padding slots are zero."""
import numpy as np
import torch

from lzzx_nerf_amd import ngp
from oracle import ngp as ONGP
from oracle.head import korder_chained


def test_fragment_tables_match_the_checkers_summation_orders():
    layer, row, col, keep = ngp._fragment_tables()
    assert layer.shape == (ngp.NGP_FRAGS * 64,)
    lane = np.arange(64)
    base = {0: (0, 8, 4), 1: (32, 16, 1), 2: (48, 8, 4), 3: (80, 16, 1)}          # first fragment, k-steps, feature tiles
    want = {0: ONGP.korder_levels(), 1: korder_chained(64), 2: ONGP.korder_color0(), 3: korder_chained(64)}
    for L, (f0, KS, NT) in base.items():
        order = []
        for ks in range(KS):
            frag = f0 + ks * NT                     # feature tile 0 of this k-step
            for q in range(4):                      # a 16x16x4 MFMA sums k = 0..3 = the four lane groups in order
                i = frag * 64 + 16 * q              # lane (m = 0, q)
                assert layer[i] == L
                order.append(int(col[i]) if keep[i] else -1)
        assert order == [int(k) for k in want[L]], L
        # every feature tile of a k-step reads the same input column; its rows are 16 ft + m
        for ks in range(KS):
            for ft in range(NT):
                sl = slice((f0 + ks * NT + ft) * 64, (f0 + ks * NT + ft + 1) * 64)
                assert np.array_equal(col[sl][keep[sl]], col[(f0 + ks * NT) * 64: (f0 + ks * NT + 1) * 64][keep[sl]])
                if L in (0, 2):
                    assert np.array_equal(row[sl], 16 * ft + (lane & 15))


def test_pack_weights_places_every_weight_once_and_zero_pads():
    g = torch.Generator().manual_seed(0)
    ws = [torch.rand(64, 32, generator=g) + 1, torch.rand(16, 64, generator=g) + 1, torch.rand(64, 31, generator=g) + 1, torch.rand(3, 64, generator=g) + 1]
    packed = ngp.pack_weights(*ws).numpy()
    layer, row, col, keep = ngp._fragment_tables()
    assert packed.shape == (ngp.NGP_FRAGS * 64,) and int(keep.sum()) == sum(w.numel() for w in ws)
    assert np.all(packed[~keep] == 0.0)                                          # padding: colour_net.1 rows 3..15, the sigma slot of colour_net.0
    for L, w in enumerate(ws):
        sel = (layer == L) & keep
        seen = np.zeros(tuple(w.shape), int)
        np.add.at(seen, (row[sel], col[sel]), 1)
        assert np.all(seen == 1)
        assert np.array_equal(packed[sel], w.numpy()[row[sel], col[sel]])
    # the colour net never sees sigma_net's row 0 (sigma's pre-activation): its slot is masked, the geometry features follow at columns 16..30
    c0 = (layer == 2)
    assert int((~keep[c0]).sum()) == 4 * 16 and set(col[c0 & keep]) == set(range(31))
