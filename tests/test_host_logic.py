"""Host-side logic that needs no GPU: copy_vbmf_params for all four parameter types (examples/mil_util.jl:212-290), the
content fingerprint of the device cache, the default device dtype of the reference-style API."""
import numpy as np
import pytest

import __graft_entry__ as G


@pytest.fixture(scope="module")
def pkg():
    return G.load_package()


def test_copy_vbmf_params_all_types(pkg):
    rng = np.random.default_rng(0)
    Y, Ynew = rng.standard_normal((12, 9)), rng.standard_normal((12, 5))      # same rows (the basis), another M
    H, H0 = 4, 2
    # basic
    p = pkg.vbmf_init(Y, H, sigma2=0.3, H1=1, labels=[2], rng=rng)
    q = pkg.copy_vbmf_params(Ynew, p, rng=rng)
    assert (q.L, q.M, q.H, q.H1) == (12, 5, H, 0) and q.labels.size == 0 and q.sigma2 == 0.3
    assert np.array_equal(q.BHat, p.BHat) and q.BHat is not p.BHat and np.array_equal(q.invCB, p.invCB)
    # sparse
    s = pkg.vbmf_sparse_init(Y, H, gamma0=1e-3, rng=rng)
    s.delta = s.delta * 3.0
    q = pkg.copy_vbmf_params(Ynew, s, rng=rng)
    assert (q.M, q.MH) == (5, 5 * H) and q.gamma == s.gamma and np.array_equal(q.delta, s.delta) and q.gamma0 == 1e-3
    # two groups: the fitted hyper-priors travel (:257-260)
    d = pkg.vbmf_dual_init(Y, H, H0, rng=rng)
    d.alpha00, d.beta00, d.alpha01, d.beta01 = 0.1, 0.2, 0.3, 0.4
    q = pkg.copy_vbmf_params(Ynew, d, rng=rng)
    assert isinstance(q, pkg.vbmf_dual_parameters) and (q.M, q.H0, q.H1) == (5, H0, H - H0)
    assert (q.alpha00, q.beta00, q.alpha01, q.beta01) == (0.1, 0.2, 0.3, 0.4) and np.array_equal(q.BHat, d.BHat)
    # three groups: TWO sets, the second with group 3's priors in group 2's place, M0 = M, third group at 1e-10 (:262-290)
    t = pkg.vbmf_trial_init(Y, H, H0, 4, rng=rng)
    t.alpha01, t.beta01, t.alpha02, t.beta02, t.alpha03, t.beta03 = 0.1, 0.2, 0.3, 0.4, 0.5, 0.6
    q0, q1 = pkg.copy_vbmf_params(Ynew, t, rng=rng)
    for q in (q0, q1):
        assert isinstance(q, pkg.vbmf_trial_parameters) and (q.M, q.M0, q.M1) == (5, 5, 0) and np.array_equal(q.CB, t.CB)
        assert (q.alpha01, q.beta01, q.alpha03, q.beta03) == (0.1, 0.2, 1e-10, 1e-10)
    assert (q0.alpha02, q0.beta02) == (0.3, 0.4) and (q1.alpha02, q1.beta02) == (0.5, 0.6)


def test_fingerprint_and_default_dtype(pkg):
    assert pkg._defaults["y_dtype"] == pkg.VBMF_Y_F32          # fp32 storage of the caller's Float64 Y unless opted out
    rng = np.random.default_rng(1)
    for order in ("C", "F"):
        Y = np.asarray(rng.standard_normal((300, 200)), order=order)
        f0 = pkg._fingerprint(Y)
        assert pkg._fingerprint(Y) == f0
        Y[17, 3] += 1e-9
        f1 = pkg._fingerprint(Y)
        Y *= 2.0
        assert len({f0, f1, pkg._fingerprint(Y)}) == 3
    big = np.zeros((4096, 2048))                                # sampled beyond 4M elements: a scaling is still seen
    f0 = pkg._fingerprint(big)
    big += 1.0
    assert pkg._fingerprint(big) != f0


def test_parity_baseline_guard():
    """The measured-error baseline under every GPU comparison (tests/helpers.py): within 3x of the measured figure passes,
    a tenfold regression on one field fails, unknown tags and non-comparison lines are ignored."""
    import json
    import os
    import pytest
    from tests import helpers
    base = json.load(open(helpers.BASELINE))
    tag = "f32 777x555 H64 run3"
    assert tag in base and base[tag]["BHat"] > 0
    ok = f"{tag}: " + " ".join(f"{k}={2.5 * v:.2e}" for k, v in base[tag].items())
    helpers.baseline_guard(ok)
    bad = f"{tag}: " + " ".join(f"{k}={(10 * v if k == 'SigmaB' else v):.2e}" for k, v in base[tag].items())
    with pytest.raises(AssertionError, match="SigmaB"):
        helpers.baseline_guard(bad)
    helpers.baseline_guard("no such tag: A=1.00e+00")
    helpers.baseline_guard("   d trace: max abs dev 4.76e-07 at sweep 0")
    tiny = f"{tag}: SigmaA={helpers.BASELINE_FLOOR * 0.9:.2e}"          # below the noise floor: never a regression
    helpers.baseline_guard(tiny)
