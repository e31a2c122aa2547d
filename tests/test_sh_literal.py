"""Known-answer test of the SH encoder against the reference's LITERAL polynomials and Jacobian tables
(/root/reference/shencoder/src/shencoder.cu:49-121, :130-350), evaluated from the reference's text in the build container by
tests/golden/make_golden_sh.py (values only are stored).  The product evaluates another factorisation (include/lzzx_sh_eval.h); the GPU
kernel is bit-identical to the checker (tests/test_gpu_parity.py::test_sh_bit_exact), so pinning the checker pins both.

Tolerances, in ulps of each output column's largest magnitude:
  * degree <= 4 (the path uses degree 4, network.py:147): <= 4 ulp against the reference's own float32 evaluation;
  * degrees 5..8: the literal polynomials cancel heavily in float32 (their own float32 evaluation is up to ~33 ulp from their exact
    value), so the statement is: the checker is at least as close to the exact polynomial (the same text evaluated in float64) as the
    reference's float32 evaluation is, plus 4 ulp."""
import os

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _directions(seed, n):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    d[:7] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0.57735026, 0.57735026, 0.57735026]]
    return d


@pytest.fixture(scope="module")
def lit():
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_sh_literal.npz"))
    seed, n = [int(v) for v in g["seed"]]
    return g, _directions(seed, n)


def _ulps(a, b, scale_from):
    scale = np.maximum(np.abs(scale_from).max(0), 1e-30).astype(np.float32)
    return np.abs(a.astype(np.float64) - b) / np.spacing(scale)


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
def test_sh_matches_reference_literals_to_4_ulp(lit, degree):
    g, d = lit
    n = degree * degree
    out, jac = O.sh_encode_forward(d, degree, True)
    jac = jac.reshape(len(d), 3, n)
    assert _ulps(out, g["values"][:, :n], g["values"][:, :n]).max() <= 4
    assert _ulps(jac, g["jacobian"][:, :, :n], g["jacobian"][:, :, :n]).max() <= 4


@pytest.mark.parametrize("degree", [5, 6, 7, 8])
def test_sh_high_degrees_at_least_as_exact_as_the_literal_float32_evaluation(lit, degree):
    g, d = lit
    lo, n = (degree - 1) ** 2, degree * degree
    out, jac = O.sh_encode_forward(d, degree, True)
    jac = jac.reshape(len(d), 3, n)
    ours = _ulps(out[:, lo:n], g["values_f64"][:, lo:n], g["values_f64"][:, lo:n]).max()
    theirs = _ulps(g["values"][:, lo:n], g["values_f64"][:, lo:n], g["values_f64"][:, lo:n]).max()
    assert ours <= theirs + 4, (ours, theirs)
    ours = _ulps(jac[:, :, lo:n], g["jacobian_f64"][:, :, lo:n], g["jacobian_f64"][:, :, lo:n]).max()
    theirs = _ulps(g["jacobian"][:, :, lo:n], g["jacobian_f64"][:, :, lo:n], g["jacobian_f64"][:, :, lo:n]).max()
    assert ours <= theirs + 4, (ours, theirs)
    # and in absolute terms both sit within 64 ulp of each other
    assert _ulps(out[:, lo:n], g["values"][:, lo:n], g["values"][:, lo:n]).max() <= 64
