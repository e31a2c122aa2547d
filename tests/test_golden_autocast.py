"""The checker's half-precision head (oracle.head.head_forward_fp16: what cfg5 and every f16 leg of the HIP path are graded against)
held to the REFERENCE's own NeRFNetwork.forward run under torch autocast(float16) -- tests/golden/reference_autocast.npz, produced by
tests/golden/make_golden_autocast.py from the imported reference Python (network.py:252-311).

The fixture ran under CPU autocast (no CUDA device in the build container).  For this graph CPU and CUDA autocast apply the same policy
to every op but three; the checker implements the CUDA policy (that is what the reference runs: torch.cuda.amp.autocast,
TrainerUtil.py:455,535,649,858), so those three are compared THROUGH the half value both policies share:

  op (network.py)                  CUDA list (torch/amp docs, "CUDA ops that autocast to float32")   CPU autocast    how it is checked here
  torch.exp(h[..., 0])      :302   exp  -> f32 in, f32 out                                           half exp        pre-activation (sigma_net.net.2
                                                                                                                     column 0) to half rounding; then
                                                                                                                     half(checker sigma) vs the fixture
  aud_ch_att.norm(dim=-1)   :308   norm -> f32                                                       half norm       att (aud_ch_att_net.net.1) to half
                                                                                                                     rounding; half(checker norm) vs fixture
  log(1 + exp(unc))         :278   exp, log -> f32                                                   half throughout unc_net.net.1 output to half rounding;
                                                                                                                     the half sequence replayed in numpy
Everything else -- Linear (half list on both), cat (promote on both), relu / sigmoid / mul / sub (unlisted: input type, ordinary type
promotion) -- is compared directly.  The recorded aten op trace is asserted below so the table above cannot rot silently.

Tolerance: a half Linear fixes no summation order (mkldnn / cpublas here, numpy's f32 matmul in the checker, the matrix core in the
kernel), so two correct implementations can differ by ONE half ulp wherever the f32 sum lies next to a rounding boundary, and a
flipped half can move the next layer by a few more.  Bounds: `HALF_ULPS` ulps of the LAYER's largest magnitude class, measured per
element as |a - b| <= ulps * ulp_half(max(|a|, |b|)) with a floor of one ulp at 2^-10 (values near zero are sums with cancellation).
"""
import os

import numpy as np
import pytest

from oracle.head import TriplaneSpec, head_forward_fp16

HERE = os.path.dirname(os.path.abspath(__file__))
F16, F32 = np.float16, np.float32
LAYERS = ["aud_ch_att_net.net.0", "aud_ch_att_net.net.1", "eye_att_net.net.0", "eye_att_net.net.1", "sigma_net.net.0", "sigma_net.net.1",
          "sigma_net.net.2", "color_net.net.0", "color_net.net.1"]


@pytest.fixture(scope="module")
def ac():
    return np.load(os.path.join(HERE, "golden", "reference_autocast.npz"), allow_pickle=False)


def half_ulp(x):
    """spacing of float16 at |x| (normal range), never below the spacing at 2^-10"""
    e = np.floor(np.log2(np.maximum(np.abs(x.astype(np.float64)), 2.0 ** -10)))
    return 2.0 ** (e - 10)


def ulps_apart(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / half_ulp(np.maximum(np.abs(a), np.abs(b)))


def layer(ac, tag, name):
    v = ac[f"{tag}_lin/{name}"]
    return ac[f"h_test_lin/{name}"] if v.dtype.kind == "U" else v       # "=h_test": the arrangement reproduced h_test's bits


def run(golden, params, tag, testing):
    tr = {}
    out = head_forward_fp16(TriplaneSpec(1.0), params, golden["net_xyz"], golden["net_dirs"], golden["net_enc_a"], golden["net_ind"],
                            golden["net_eye"], testing=testing, enc_a_half=tag == "h", trace=tr)
    return out, tr


@pytest.mark.parametrize("tag", ["h", "f"])
def test_every_linear_output_matches_reference_autocast(golden, params, ac, tag):
    (_, _, _, _, _), tr = run(golden, params, tag, testing=False)
    worst = {}
    for name in LAYERS + ["unc_net.net.0", "unc_net.net.1"]:
        ref = layer(ac, f"{tag}_train", name)
        got = tr[name][:ref.shape[0]]
        assert got.dtype == F16 and ref.dtype == F16 and got.shape == ref.shape, name
        u = ulps_apart(got, ref)
        worst[name] = (float(u.max()), float((got == ref).mean()))
        # first layers see identical inputs: at most one ulp and almost always none; deeper layers inherit flipped halves
        depth = int(name[-1])
        assert u.max() <= (1.0 if depth == 0 else 4.0 if name != "sigma_net.net.2" else 8.0), (name, worst[name])
        assert (got == ref).mean() >= (0.99 if depth == 0 else 0.90), (name, worst[name])
    print(worst)


@pytest.mark.parametrize("tag", ["h", "f"])
def test_outputs_match_reference_autocast(golden, params, ac, tag):
    (sig, rgb, aa, ae, unc), tr = run(golden, params, tag, testing=True)
    p = f"{tag}_test_"
    # ---- ops both policies treat alike: compared directly ------------------------------------------------------------------------
    assert str(ac[p + "rgb_dtype"]) == "float16" and str(ac[p + "amb_eye_dtype"]) == "float16"
    assert ulps_apart(rgb, ac[p + "rgb"]).max() <= 4 and (rgb.astype(F16) == ac[p + "rgb"]).mean() > 0.9
    assert np.array_equal(rgb.astype(F16).astype(F32), rgb), "the checker's rgb is a half value (sigmoid * 1.002 - 0.001 in half)"
    assert ulps_apart(ae, ac[p + "amb_eye"]).max() <= 2 and (ae.astype(F16) == ac[p + "amb_eye"]).mean() > 0.95
    # ---- exp: CUDA list says f32; the fixture's CPU exp rounded to half -----------------------------------------------------------
    assert str(ac[p + "sigma_dtype"]) == "float16" and sig.dtype == F32
    pre = tr["sigma_net.net.2"][:, 0].astype(F32)
    assert np.array_equal(sig, _exp32(pre)) or np.allclose(sig, np.exp(pre.astype(np.float64)), rtol=3e-7), "checker: f32 exp of the half pre-activation"
    # e^x moves by x-ulps * |x| * e^x: a pre-activation that is k half ulps off moves sigma by ~ k * |x| * 2^-10 relative
    ref_pre = layer(ac, f"{tag}_test", "sigma_net.net.2")[:, 0]
    n = ref_pre.shape[0]
    k = ulps_apart(pre[:n], ref_pre).max()
    ref_sig = ac[p + "sigma"][:n].astype(np.float64)
    rel = np.abs(sig[:n].astype(np.float64) - ref_sig) / ref_sig
    assert rel.max() <= (k * max(np.abs(pre[:n]).max(), 1.0) + 1.0) * 2.0 ** -10, (rel.max(), k)
    same_pre = pre[:n].astype(F16) == ref_pre
    assert same_pre.mean() > 0.8
    assert ulps_apart(sig[:n][same_pre].astype(F16), ac[p + "sigma"][:n][same_pre]).max() <= 1, "same half pre-activation => same half sigma (+- exp rounding)"
    # ---- norm: CUDA list says f32 ---------------------------------------------------------------------------------------------------
    assert str(ac[p + "amb_aud_dtype"]) == "float16" and aa.dtype == F32
    assert ulps_apart(aa.astype(F16), ac[p + "amb_aud"]).max() <= 2
    att = tr["aud_ch_att_net.net.1"].astype(np.float64)
    assert np.allclose(aa[:, 0], np.sqrt((att ** 2).sum(1)), rtol=2e-7), "checker: f32 norm of the half att"
    # ---- test mode: zeros_like(enc_x) is f32 in both, log(1 + exp(0)) = ln 2 --------------------------------------------------------
    assert str(ac[p + "unc_dtype"]) == "float32" and np.allclose(unc, ac[p + "unc"], atol=1e-7)


def _exp32(x):
    from oracle import oracle as O
    return O.unary("exp", np.ascontiguousarray(x, dtype=F32))


@pytest.mark.parametrize("tag", ["h", "f"])
def test_training_mode_uncertainty(golden, params, ac, tag):
    (_, _, _, _, unc), tr = run(golden, params, tag, testing=False)
    ref_u2 = layer(ac, f"{tag}_train", "unc_net.net.1")
    u2 = tr["unc_net.net.1"]
    n = ref_u2.shape[0]
    assert ulps_apart(u2[:n], ref_u2).max() <= 4
    # the fixture's CPU policy: exp, 1 +, log all in half.  Replay it from the fixture's own pre-activation ...
    with np.errstate(over="ignore"):
        replay = np.log((F16(1) + np.exp(ref_u2.astype(F32)).astype(F16)).astype(F32)).astype(F16)
    got = ac[f"{tag}_train_unc"].reshape(-1, 1)[:n]
    assert str(ac[f"{tag}_train_unc_dtype"]) == "float16" and ulps_apart(replay, got).max() <= 1
    # ... and the CUDA policy (exp / log on the fp32 list) from the checker's: f32 softplus of the half pre-activation
    assert unc.dtype == F32 and np.allclose(unc, np.log1p(np.exp(u2.astype(np.float64))), rtol=3e-6, atol=1e-7)
    # both agree to what half can hold of log(1 + e^u)
    assert np.abs(unc[:n].astype(np.float64) - got.astype(np.float64)).max() <= 3e-3


def test_density_entry_point(golden, params, ac):
    """NeRFNetwork.density under autocast (renderer.py:744, the occupancy-grid update) = the same sigma / geo as forward"""
    (sig, _, _, _, _), tr = run(golden, params, "h", testing=True)
    assert np.array_equal(ac["h_density_sigma"], ac["h_test_sigma"])
    n = ac["h_density_geo"].shape[0]
    assert np.array_equal(ac["h_density_geo"], ac["h_test_lin/sigma_net.net.2"][:, 1:])
    assert ulps_apart(tr["sigma_net.net.2"][:n, 1:], ac["h_density_geo"]).max() <= 8


def test_recorded_policy(ac):
    """the aten trace the reference's forward produced under autocast: which op ran in which type"""
    tr = [str(r) for r in ac["h_train_op_trace"]]
    mm = [r for r in tr if r.startswith("mm.")]
    assert len(mm) == 11 and all(r == "mm.default(float16,float16)->float16" for r in mm)            # 11 Linear layers, all half
    assert tr.count("relu_.default(float16)->float16") == 6                                          # in-place half ReLU, network.py:89
    assert tr.count("sigmoid.default(float16)->float16") == 2
    assert "mul.Tensor(float16,float16)->float16" in tr                                              # enc_a (half) * att
    assert "mul.Tensor(float32,float16)->float32" in tr                                              # e (f32 [1,1]) * eye_att: type promotion
    cats = [r for r in tr if r.startswith("cat.")]
    assert cats[-2:] == ["cat.default(float32,float32,float32)->float32"] * 2                        # promote to widest: [enc_x|enc_w|e], [enc_d|geo|c]
    # the three ops CPU autocast leaves in half and CUDA autocast sends to f32
    assert tr.count("exp.default(float16)->float16") == 2 and "log.default(float16)->float16" in tr
    assert "linalg_vector_norm.default(float16)->float16" in tr
    trf = [str(r) for r in ac["f_train_op_trace"]]
    assert "mul.Tensor(float32,float16)->float32" in trf and "mul.Tensor(float16,float16)->float16" not in trf
    for name in LAYERS:
        want = "float32" if name in ("aud_ch_att_net.net.0", "eye_att_net.net.0", "sigma_net.net.0", "color_net.net.0") else "float16"
        assert str(ac[f"h_test_lin_in_dtype/{name}"]) == want, name        # first layers are handed f32 (enc_x / the promoted cat) and cast it
