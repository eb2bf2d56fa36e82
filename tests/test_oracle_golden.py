"""Pin the CPU oracle against the reference's own recorded trajectories (tests/golden/*.npz).

These are the only known-answer vectors the reference holds for this path (its test/runtests.jl is
the always-failing package template).  Tolerance 1e-10 relative to max|field| (fp64 replay; the
recorded run used LU `inv`, same as numpy).
"""
import os

import numpy as np
import pytest

from oracle import vbmf_oracle as O

TOL = 1e-10


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _load_basic(golden_dir, t):
    g = np.load(os.path.join(golden_dir, "vbmf_test.npz"))
    p = O.vbmf_parameters()
    p.L, p.M, p.H, p.H1 = int(g["L"][t]), int(g["M"][t]), int(g["H"][t]), int(g["H1"][t])
    for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB"):
        setattr(p, f, g[f][t].copy())
    p.sigma2 = float(g["sigma2"][t])
    return g, p


@pytest.mark.parametrize("fused", [False, True])
def test_basic_trajectory_replay(golden_dir, fused):
    g, p = _load_basic(golden_dir, 0)
    Y = g["Y"]
    assert Y.shape == (10, 20) and abs(O.norm2(Y) - 189.80687013285842) < 1e-11
    worst = {}
    for t in range(1, 101):
        O.vbmf_(Y, p, 1, eps=0.0, est_covs=True, est_var=True, fused=fused)
        for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB"):
            worst[f] = max(worst.get(f, 0.0), _rel(getattr(p, f), g[f][t]))
        worst["sigma2"] = max(worst.get("sigma2", 0.0), abs(p.sigma2 - g["sigma2"][t]) / g["sigma2"][t])
    assert max(worst.values()) < TOL, worst


def test_basic_known_answers(golden_dir):
    """Spot values quoted in SURVEY.md Appendix B / BASELINE.md section 2."""
    g, p = _load_basic(golden_dir, 0)
    Y = g["Y"]
    tr = []
    O.vbmf_(Y, p, 100, eps=0.0, est_covs=True, est_var=True, trace=tr)
    s2 = [t[1] for t in tr]
    assert abs(s2[0] - 0.6259945822471235) < 1e-12
    assert abs(s2[1] - 0.3115297322579878) < 1e-12
    assert abs(s2[2] - 0.20191822567017795) < 1e-12
    assert abs(s2[99] - 0.0023457154169626905) < 1e-12
    assert np.allclose(np.diag(p.CA), [0.0839409657264668, 0.17933104854249735], rtol=1e-10)
    # spectral d (derived values in Appendix B, 5 significant digits)
    d = [t[0] for t in tr]
    assert abs(d[0] - 1.1591) < 1e-4 and abs(d[1] - 0.44930) < 1e-5
    assert abs(d[9] - 1.1545e-4) < 1e-8 and abs(d[99] - 3.1045e-6) < 1e-10
    # all 100 sweeps ran in the recording => d never <= 1e-6
    assert min(d) > 1e-6


def test_each_basic_update_from_recorded_state(golden_dir):
    """One sweep from EVERY recorded slice reproduces the next slice (no error accumulation)."""
    g = np.load(os.path.join(golden_dir, "vbmf_test.npz"))
    Y = g["Y"]
    for t in range(0, 100, 7):
        _, p = _load_basic(golden_dir, t)
        O.updateA(Y, p)
        assert _rel(p.SigmaA, g["SigmaA"][t + 1]) < TOL and _rel(p.AHat, g["AHat"][t + 1]) < TOL
        O.updateB(Y, p)
        assert _rel(p.SigmaB, g["SigmaB"][t + 1]) < TOL and _rel(p.BHat, g["BHat"][t + 1]) < TOL
        O.updateCA(p); O.updateCB(p)
        assert _rel(p.invCA, g["invCA"][t + 1]) < TOL and _rel(p.invCB, g["invCB"][t + 1]) < TOL
        O.updateSigma2(Y, p)
        assert abs(p.sigma2 - g["sigma2"][t + 1]) / g["sigma2"][t + 1] < TOL


def _load_sparse(golden_dir, t):
    g = np.load(os.path.join(golden_dir, "sparse_test.npz"))
    p = O.vbmf_sparse_parameters()
    p.L, p.M, p.H, p.H1 = int(g["L"][t]), int(g["M"][t]), int(g["H"][t]), int(g["H1"][t])
    p.MH = p.M * p.H
    for f in ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta"):
        setattr(p, f, g[f][t].copy())
    for f in ("alpha0", "beta0", "alpha", "gamma0", "delta0", "gamma", "sigmaHat", "eta0", "zeta0", "eta", "zeta",
              "trYTY"):
        setattr(p, f, float(g[f][t]))
    return g, p


def test_sparse_fullcov_trajectory_replay(golden_dir):
    g, p = _load_sparse(golden_dir, 0)
    Y = g["Y"]
    assert abs(p.alpha - 0.5000000001) < 1e-15 and abs(p.gamma - 5.0000000001) < 1e-12
    worst = {}
    cov = {int(s): i for i, s in enumerate(g["cov_slices"])}
    for t in range(1, 101):
        O.vbmf_sparse_(Y, p, 1, eps=0.0, full_cov=True, est_cb=True)
        for f in ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta"):
            worst[f] = max(worst.get(f, 0.0), _rel(getattr(p, f), g[f][t]))
        for f in ("sigmaHat", "zeta"):
            worst[f] = max(worst.get(f, 0.0), abs(getattr(p, f) - g[f][t]) / abs(g[f][t]))
        if t in cov:
            worst["SigmaATVec"] = max(worst.get("SigmaATVec", 0.0), _rel(p.SigmaATVec, g["SigmaATVec"][cov[t]]))
            worst["invSigmaATVec"] = max(worst.get("invSigmaATVec", 0.0),
                                         _rel(p.invSigmaATVec, g["invSigmaATVec"][cov[t]]))
    assert max(worst.values()) < TOL, worst
    assert abs(p.sigmaHat - 4.60278260971759) < 1e-10
    assert abs(p.zeta - 21.72598805535064) < 1e-9


def test_sparse_diag_equals_full_when_H1(golden_dir):
    """Unpinned diagonal branch: with H=1 the full covariance IS diagonal, and the QS1 layout quirk
    vanishes (a single v); QS2 (sigmaHat not multiplying L*SigmaB) is the only difference, so compare
    from a state with SigmaB = 0."""
    rng = np.random.default_rng(3)
    Y, _, _ = O.toy_matrix(12, 9, 1, 0.05, rng)
    p1 = O.vbmf_sparse_init(Y, 1, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(4))
    p2 = O.vbmf_sparse_init(Y, 1, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(4))
    O.sparse_updateA(Y, p1, full_cov=True)
    O.sparse_updateA(Y, p2, full_cov=False)
    assert _rel(p2.ATVecHat, p1.ATVecHat) < 1e-12 and _rel(p2.diagSigmaATVec, p1.diagSigmaATVec) < 1e-12
    assert _rel(p2.SigmaA, p1.SigmaA) < 1e-12


def test_spread_v_quirk():
    v = np.array([1.0, 2.0, 3.0])
    assert O.spread_v(v, 3, True).tolist() == [1, 2, 3, 1, 1, 2, 2, 3, 3]
    assert O.spread_v(v, 3, False).tolist() == [1, 2, 3, 1, 2, 3, 1, 2, 3]


def test_lowerbound_value_on_final_sparse_state(golden_dir):
    """Not a reference output (none recorded) -- regression value of the restatement, SURVEY 8(c)."""
    g, p = _load_sparse(golden_dir, 100)
    lb = O.lowerBound(g["Y"], p)
    assert abs(lb - (-1183.5705028312561)) < 1e-6


def test_elbo_basic_monotone_fixed_hyper():
    """Property test for the build-defined ELBO: non-decreasing with est_covs=est_var=False."""
    rng = np.random.default_rng(11)
    Y, _, _ = O.toy_matrix(40, 30, 3, 0.05, rng)
    p = O.vbmf_init(Y, 3, ca=0.5, cb=0.5, sigma2=0.05, rng=rng)
    tr = []
    O.vbmf_(Y, p, 30, eps=0.0, trace=tr)
    e = np.array([t[2] for t in tr])
    assert np.all(np.isfinite(e)) and np.all(np.diff(e) > -1e-8 * np.abs(e[:-1]))


def test_fused_equals_faithful_with_mask():
    rng = np.random.default_rng(5)
    Y, _, _ = O.toy_matrix(50, 40, 4, 0.05, rng)
    kw = dict(ca=0.1, cb=0.1, sigma2=0.1, H1=2, labels=[0, 3, 17, 39])
    p1 = O.vbmf_init(Y, 4, rng=np.random.default_rng(6), **kw)
    p2 = O.vbmf_init(Y, 4, rng=np.random.default_rng(6), **kw)
    assert np.all(p1.AHat[[0, 3, 17, 39], 2:] == 0)
    _, n1, d1 = O.vbmf_(Y, p1, 25, eps=0.0, est_covs=True, est_var=True, fused=False)
    _, n2, d2 = O.vbmf_(Y, p2, 25, eps=0.0, est_covs=True, est_var=True, fused=True)
    for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB"):
        assert _rel(getattr(p2, f), getattr(p1, f)) < 1e-10
    assert np.all(p1.AHat[[0, 3, 17, 39], 2:] == 0)
    assert abs(p1.sigma2 - p2.sigma2) / p1.sigma2 < 1e-10 and abs(d1 - d2) / d1 < 1e-8


def test_termination_semantics():
    """src/vbmf.jl:193,221 -- loop stops when d <= eps; reported count is i-1; NaN d exits."""
    rng = np.random.default_rng(8)
    Y, _, _ = O.toy_matrix(30, 20, 2, 0.05, rng)
    p = O.vbmf_init(Y, 2, ca=0.1, cb=0.1, sigma2=0.1, rng=rng)
    _, n, d = O.vbmf_(Y, p, 500, eps=1e-3, est_covs=True, est_var=True)
    assert n < 500 and d <= 1e-3
    p = O.vbmf_init(Y, 2, rng=rng)
    _, n, d = O.vbmf_(Y, p, 0)
    assert n == 0
