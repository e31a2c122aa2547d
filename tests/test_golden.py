"""CPU checker vs vectors produced by the reference's own Python (tests/golden/make_golden.py, run in the
build container with /root/reference on the path).  These pin layouts, concatenation order, activations and
the ray convention; float agreement is to torch-CPU rounding (the summation order of a torch Linear is free)."""
import numpy as np

from oracle import oracle as O
from oracle.head import TriplaneSpec, density, encode_x, get_rays, head_forward, korder_chained, korder_natural


def test_table_layouts(golden):
    spec = TriplaneSpec(1.0)
    assert np.array_equal(spec.offsets, golden["layout_triplane_offsets"])
    assert spec.per_level_scale == golden["layout_triplane_pls"][0]
    assert list(golden["layout_triplane_shape"]) == [163584, 1] and golden["layout_triplane_outdim"][0] == 12
    # offsets [0, 4232, 10480, 19512, 32512, 48896, ... +16384 ..., 163584]
    assert list(spec.offsets[:6]) == [0, 4232, 10480, 19512, 32512, 48896] and np.all(np.diff(spec.offsets[5:]) == 16384)
    pls = golden["layout_hashgrid_default_pls"][0]
    assert np.array_equal(O.grid_offsets(3, 16, pls, 16, 19), golden["layout_hashgrid_default_offsets"])
    assert list(golden["layout_hashgrid_default_shape"]) == [6119864, 2]
    pls = golden["layout_tiled_torso_pls"][0]
    assert np.array_equal(O.grid_offsets(2, 16, pls, 16, 16), golden["layout_tiled_torso_offsets"])
    # package-side restatement of the same layout rule
    from lzzx_nerf_amd.gridencoder import grid_offsets
    assert grid_offsets(2, 12, spec.per_level_scale, 64, 14) == list(golden["layout_triplane_offsets"])
    assert grid_offsets(3, 16, golden["layout_hashgrid_default_pls"][0], 16, 19) == list(golden["layout_hashgrid_default_offsets"])


def test_state_dict_contract(golden):
    keys = list(golden["sd_keys"])
    shapes = dict(zip(keys, golden["sd_shapes"]))
    for k, s in {"encoder_xy.embeddings": "(163584, 1)", "encoder_yz.offsets": "(13,)", "density_bitfield": "(262144,)",
                 "density_grid": "(1, 2097152)", "step_counter": "(16, 2)", "sigma_net.net.0.weight": "(64, 69)",
                 "sigma_net.net.2.weight": "(65, 64)", "color_net.net.0.weight": "(64, 84)", "color_net.net.1.weight": "(3, 64)",
                 "aud_ch_att_net.net.1.weight": "(32, 64)", "eye_att_net.net.1.weight": "(1, 16)", "unc_net.net.0.weight": "(32, 36)",
                 "individual_codes": "(10, 4)", "aabb_infer": "(6,)"}.items():
        assert shapes[k] == s, k


def test_encode_x_matches_reference_python(golden, params):
    spec = TriplaneSpec(1.0)
    enc_x = encode_x(spec, golden["net_xyz"], params)
    # the fixture's enc_x went through GridEncoder.forward (map to [0,1], [L,B,C] -> [B,L*C] permute) and torch.cat
    assert np.array_equal(enc_x, golden["net_enc_x"])
    assert np.all(enc_x[3, :12] == 0) and np.all(enc_x[3, 24:] == 0) and np.any(enc_x[3, 12:24] != 0)  # x = 1.5 out of range


def test_linear_orders_agree_and_match_torch(golden, params):
    enc_x = golden["net_enc_x"]
    W0, W1 = params["aud_ch_att_net.net.0.weight"], params["aud_ch_att_net.net.1.weight"]
    a_nat = O.linear(enc_x, W0, None, relu=True)
    a_k = O.linear(enc_x, W0, korder_natural(36), relu=True)
    assert np.array_equal(a_nat, a_k)
    assert np.allclose(a_nat, golden["net_aud_hidden"], atol=2e-6, rtol=1e-5)
    att_nat = O.linear(a_nat, W1)
    att_ch = O.linear(a_nat, W1, korder_chained(64))
    assert np.allclose(att_nat, att_ch, atol=2e-6) and np.allclose(att_ch, golden["net_att"], atol=3e-6, rtol=1e-5)
    assert sorted(k for k in korder_chained(64)) == list(range(64))
    assert korder_chained(16)[:8] == [0, 4, 8, 12, 1, 5, 9, 13]


def test_head_matches_reference_python(golden, params):
    spec = TriplaneSpec(1.0)
    sig, rgb, aa, ae, unc = head_forward(spec, params, golden["net_xyz"], golden["net_dirs"], golden["net_enc_a"], golden["net_ind"],
                                         golden["net_eye"], testing=True)
    assert np.allclose(sig, golden["net_sigma"], rtol=2e-5, atol=1e-6)
    assert np.allclose(rgb, golden["net_rgb"], atol=2e-6)
    assert np.allclose(aa, golden["net_amb_aud"], rtol=1e-5, atol=1e-6)
    assert np.allclose(ae, golden["net_amb_eye"], atol=1e-6)
    # test mode: ln 2 everywhere; the reference's tensor is over-sized [M, 36, 1] (SURVEY 8a' note 16)
    assert list(golden["net_unc_test_shape"]) == [777, 36, 1]
    assert np.allclose(unc, golden["net_unc_test_first"], atol=1e-7) and abs(unc[0, 0] - np.log(2)) < 1e-7
    *_, unc_tr = head_forward(spec, params, golden["net_xyz"], golden["net_dirs"], golden["net_enc_a"], golden["net_ind"],
                              golden["net_eye"], testing=False)
    assert golden["net_unc_train"].shape == (777, 1, 1)  # `uncertainty[..., None]`, network.py:280
    assert np.allclose(unc_tr, golden["net_unc_train"].reshape(777, 1), atol=2e-6)
    d = density(spec, params, golden["net_enc_x"], golden["net_enc_a"], golden["net_eye"])
    assert np.allclose(d["geo_feat"], golden["net_geo"], atol=3e-6, rtol=1e-5)


def test_get_rays_matches_reference_python(golden):
    for tag in ("id", "rot"):
        ro, rd = get_rays(golden[f"rays_{tag}_64_pose"], golden[f"rays_{tag}_64_intr"], 64, 64)
        assert np.array_equal(ro, golden[f"rays_{tag}_64_o"])
        assert np.max(np.abs(rd - golden[f"rays_{tag}_64_d"])) < 2e-7
        ro, rd = get_rays(golden[f"rays_{tag}_256_pose"], golden[f"rays_{tag}_256_intr"], 256, 256)
        assert np.max(np.abs(rd[::97] - golden[f"rays_{tag}_256_d_sub"])) < 2e-7
        assert np.allclose(rd.astype(np.float64).sum(0), golden[f"rays_{tag}_256_d_sum"], atol=1e-3)
    assert np.allclose(np.linalg.norm(rd, axis=1), 1, atol=1e-6)
