"""GPU parity of the two-group ARD variant (src/vbmf_dual.jl, full_cov=false, diag_var=false)
against the oracle.  PARITY UNPINNED: the reference holds no recorded vbmf_dual! run, and its hyper-prior fit calls
Roots.jl's `fzero`, a package it neither vendors nor pins -- the oracle takes the exact root of the same function on
the same bracket (oracle/vbmf_oracle.py, _dual_fit_shape).  The bodies shared with the sparse model (updateA!/B!/CB!/
Sigma!) are pinned through the sparse fixture."""
import numpy as np
import pytest

import __graft_entry__ as G
from oracle import vbmf_oracle as O
from tests.helpers import relF, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    G.build()
    return G.load_package()


def _mk(L, M, H, H0, seed):
    rng = np.random.default_rng(seed)
    Y, A, B = O.toy_matrix(L, M, H, 0.05, rng)
    Y = (B * np.linspace(1.0, 2.5, H)) @ A.T + 0.05 * rng.standard_normal((L, M))
    po = O.vbmf_dual_init(Y, H, H0, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(seed + 1), materialize_yhat=False)
    return Y, po


SCAL = ("L", "M", "H", "MH", "H0", "H1", "alpha00", "beta00", "alpha01", "beta01", "alpha0", "alpha1", "gamma0", "delta0",
        "gamma", "sigmaHat", "eta0", "zeta0", "eta", "zeta", "trYTY")
ARRS = ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "A0Hat", "A1Hat", "BHat", "SigmaB", "CA", "beta", "CA0", "CA1",
        "beta0", "beta1", "CB", "delta")


def _to_pkg(pkg, po):
    p = pkg.vbmf_dual_parameters()
    for f in SCAL:
        setattr(p, f, getattr(po, f))
    for f in ARRS:
        setattr(p, f, np.array(getattr(po, f), copy=True))
    p.alpha = np.array([po.alpha0, po.alpha1])
    return p


FIELDS = ("ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta")


def _cmp(tag, pg, po, tol, fields=FIELDS, priors_tol=None):
    errs = {f: relF(getattr(pg, f), getattr(po, f)) for f in fields if np.size(getattr(po, f))}
    errs["sigmaHat"] = abs(pg.sigmaHat - po.sigmaHat) / po.sigmaHat
    if priors_tol is not None:
        for f in ("alpha00", "beta00", "alpha01", "beta01"):
            errs[f] = abs(getattr(pg, f) - getattr(po, f)) / abs(getattr(po, f))
    report(f"dual {tag}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    bad = {k: v for k, v in errs.items() if not v <= (priors_tol if k in ("alpha00", "beta00", "alpha01", "beta01") else tol)}
    assert not bad, (tag, bad)


@pytest.mark.parametrize("L,M,H,H0", [(10, 20, 3, 1), (300, 170, 5, 3), (500, 260, 40, 16)])
def test_dual_each_update_f32(pkg, L, M, H, H0):
    Y, po = _mk(L, M, H, H0, 140 + H)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    for sweep in range(3):
        pg = _to_pkg(pkg, po)
        pkg.dual_updateA_(Yf, pg); O.dual_updateA(Yf, po)
        _cmp(f"{L}x{M} H{H}/{H0} s{sweep} updateA", pg, po, 5e-5, ("ATVecHat", "diagSigmaATVec", "SigmaA", "A0Hat", "A1Hat"))
        pg = _to_pkg(pkg, po)
        pkg.dual_updateB_(Yf, pg); O.sparse_updateB(Yf, po)
        _cmp(f"{L}x{M} H{H}/{H0} s{sweep} updateB", pg, po, 5e-5, ("BHat", "SigmaB"))
        pg = _to_pkg(pkg, po)
        pkg.dual_updateCA_and_priors_(pg, Y=Yf); O.dual_updateCA(po); O.dual_updatePriors(po)
        _cmp(f"{L}x{M} H{H}/{H0} s{sweep} updateCA+priors", pg, po, 5e-5, ("CA", "beta", "CA0", "CA1", "beta0", "beta1"),
             priors_tol=1e-5)
        assert pg.alpha0 == po.alpha0 and pg.alpha1 == po.alpha1           # what this updateCA! used (:324-325)
        pg = _to_pkg(pkg, po)
        pkg.dual_updateCB_(pg, Y=Yf); O.sparse_updateCB(po)
        _cmp(f"{L}x{M} H{H}/{H0} s{sweep} updateCB", pg, po, 5e-5, ("CB", "delta"))
        pg = _to_pkg(pkg, po)
        pkg.dual_updateSigma_(Yf, pg); O.sparse_updateSigma(Yf, po)
        _cmp(f"{L}x{M} H{H}/{H0} s{sweep} updateSigma", pg, po, 5e-4, ())
        assert abs(pg.zeta - po.zeta) / po.zeta < 5e-4


def test_dual_groups_use_their_own_priors(pkg):
    """updateCA! alone with two clearly different hyper-priors: each column group gets its own shape and rate."""
    L, M, H, H0 = 120, 70, 6, 2
    Y, po = _mk(L, M, H, H0, 5)
    po.alpha00, po.beta00, po.alpha01, po.beta01 = 0.7, 0.02, 3.5, 1.25
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg(pkg, po)
    pkg.dual_updateCA_(pg, Y=Y); O.dual_updateCA(po)
    _cmp("two priors updateCA", pg, po, 1e-6, ("CA", "beta", "CA0", "CA1", "beta0", "beta1"))
    assert pg.alpha0 == 1.2 and pg.alpha1 == 4.0
    assert (pg.alpha00, pg.beta00, pg.alpha01, pg.beta01) == (0.7, 0.02, 3.5, 1.25)       # no fit without est_priors


@pytest.mark.parametrize("mode,est_priors", [("f32", True), ("bf16x2", True), ("f32", False)])
def test_dual_run_and_lower_bound(pkg, mode, est_priors):
    L, M, H, H0 = 600, 380, 6, 4
    Y, po = _mk(L, M, H, H0, 121)
    ydt = pkg.VBMF_Y_F32 if mode == "f32" else pkg.VBMF_Y_BF16
    with pkg.capi.Context(L, M, H, y_dtype=ydt) as c:
        c.set_Y(Y)
        Ys = np.ascontiguousarray(c.get_Y())
    po.trYTY = float(np.sum(Ys * Ys))
    pkg.set_defaults(y_dtype=ydt, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg(pkg, po)
    d_gpu = pkg.vbmf_dual_(Ys, pg, 15, eps=0.0, est_cb=True, est_priors=est_priors)
    tr = []
    d_ref, n = O.vbmf_dual_(Ys, po, 15, eps=0.0, est_cb=True, est_priors=est_priors, trace=tr)
    tol = 2e-3 if mode == "f32" else 5e-3
    _cmp(f"run15 {mode} est_priors={est_priors}", pg, po, tol, priors_tol=tol)
    assert pg._last_run[0] == 15 and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6
    if est_priors:
        assert po.alpha00 != 1e-10 and po.alpha01 != 1e-10 and po.alpha00 != po.alpha01    # the fits moved the priors apart
    else:
        assert (pg.alpha00, pg.beta00, pg.alpha01, pg.beta01) == (1e-10, 1e-10, 1e-10, 1e-10)
    lb_gpu = pkg.lowerBound_dual(Ys, pg)
    lb_ref = O.lowerBound_dual(Ys, po)
    report(f"dual lowerBound {mode} est_priors={est_priors}: gpu {lb_gpu:.6f} oracle {lb_ref:.6f}")
    assert abs(lb_gpu - lb_ref) <= 2e-3 * abs(lb_ref)
    # the bound of the ORACLE's state evaluated on the device: isolates the bound from trajectory drift
    lb2 = pkg.lowerBound_dual(Ys, _to_pkg(pkg, po))
    assert abs(lb2 - lb_ref) <= 2e-5 * abs(lb_ref), (lb2, lb_ref)


def test_dual_without_prior_fits_is_the_sparse_model(pkg):
    """est_priors=false with equal priors: vbmf_dual! and vbmf_sparse! are the same iteration (bit for bit on the device)."""
    L, M, H, H0 = 400, 250, 8, 3
    Y, po = _mk(L, M, H, H0, 77)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_BF16, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg(pkg, po)
    pkg.vbmf_dual_(Y, pg, 8, eps=0.0, est_priors=False)
    ps = pkg.vbmf_sparse_parameters()
    for f in ("L", "M", "H", "MH", "gamma0", "delta0", "gamma", "sigmaHat", "eta0", "zeta0", "eta", "zeta", "trYTY"):
        setattr(ps, f, getattr(po, f))
    ps.alpha0, ps.beta0, ps.alpha = po.alpha00, po.beta00, po.alpha00 + 0.5
    for f in ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta"):
        setattr(ps, f, np.array(getattr(po, f), copy=True))
    pkg.vbmf_sparse_(Y, ps, 8, eps=0.0)
    for f in FIELDS:
        assert np.array_equal(getattr(pg, f), getattr(ps, f)), f
    assert pg.sigmaHat == ps.sigmaHat


def test_dual_fixed_basis_and_edge_groups(pkg):
    """vbls! on the two-group model (examples/mil_util.jl:190-193), and the degenerate splits H0 = 0 / H0 = H
    (an empty group's hyper-priors stay as they are)."""
    L, M, H = 260, 150, 5
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    for H0 in (2, 0, H):
        Y, po = _mk(L, M, H, H0, 31 + H0)
        Yf = Y.astype(np.float32).astype(np.float64)
        po.trYTY = float(np.sum(Yf * Yf))
        O.vbmf_dual_(Yf, po, 3, eps=0.0)                     # a state with non-trivial SigmaB / CB
        pg = _to_pkg(pkg, po)
        A = pkg.vbls_(Yf, pg, 6)
        O.vbls_dual_(Yf, po, 6)
        _cmp(f"vbls H0={H0}", pg, po, 2e-4, ("ATVecHat", "diagSigmaATVec", "SigmaA", "CA", "beta"))
        assert A is pg.AHat
        pg = _to_pkg(pkg, po)
        pkg.vbmf_dual_(Yf, pg, 4, eps=0.0, est_priors=True)
        O.vbmf_dual_(Yf, po, 4, eps=0.0, est_priors=True)
        _cmp(f"run4 H0={H0}", pg, po, 1e-3, priors_tol=1e-3)


def test_dual_argument_errors(pkg):
    L, M, H = 64, 48, 4
    Y, po = _mk(L, M, H, 2, 3)
    hyper = dict(alpha0=1e-10, beta0=1e-10, gamma0=1e-10, delta0=1e-10, eta0=1e-10, zeta0=1e-10)
    with pkg.capi.Context(L, M, H, y_dtype=pkg.VBMF_Y_F32, variant=pkg.capi.VBMF_VARIANT_DUAL_DIAG) as c:
        c.set_Y(Y)
        with pytest.raises(pkg.VbmfError):
            c.dual_set_priors(2, 1.0, 1.0, 1.0, 1.0)                      # no state yet
        c.sparse_set_state(po.ATVecHat, po.diagSigmaATVec, po.CA, po.beta, po.BHat, po.SigmaB, po.CB, po.delta, po.sigmaHat,
                           po.zeta, hyper)
        with pytest.raises(pkg.VbmfError, match="H0"):
            c.dual_set_priors(H + 1, 1.0, 1.0, 1.0, 1.0)                  # src/vbmf_dual.jl:126-128
        with pytest.raises(pkg.VbmfError):
            c.dual_set_priors(2, 0.0, 1.0, 1.0, 1.0)
        with pytest.raises(pkg.VbmfError):
            c.dual_set_priors(2, 1.0, 1.0, 1.0, 1.0, alpha0=-1.0)
        with pytest.raises(pkg.VbmfError):
            c.sparse_step(pkg.capi.SSTEP_PRIORS)                          # needs SSTEP_CA in the same call
        with pytest.raises(pkg.VbmfError):
            c.sparse_set_state(po.ATVecHat, po.diagSigmaATVec, po.CA, po.beta, po.BHat, po.SigmaB, po.CB, po.delta,
                               po.sigmaHat, po.zeta, hyper, labels0=[1], H1=1)   # the two-group model has no label mask
    with pkg.capi.Context(L, M, H, y_dtype=pkg.VBMF_Y_F32, variant=pkg.capi.VBMF_VARIANT_SPARSE_DIAG) as c:
        c.set_Y(Y)
        c.sparse_set_state(po.ATVecHat, po.diagSigmaATVec, po.CA, po.beta, po.BHat, po.SigmaB, po.CB, po.delta, po.sigmaHat,
                           po.zeta, hyper)
        with pytest.raises(pkg.VbmfError):
            c.dual_run(1)                                                  # not a two-group context
        with pytest.raises(pkg.VbmfError):
            c.sparse_step(pkg.capi.SSTEP_CA | pkg.capi.SSTEP_PRIORS)


@pytest.mark.parametrize("mode", ["f32", "bf16x2"])
def test_dual_heteroscedastic_run(pkg, mode):
    """vbmf_dual! with diag_var = true (src/vbmf_dual.jl:245-249, 263-264, 294-299, 371-378: the sparse model's
    heteroscedastic bodies) and the prior fits, 8 sweeps; then vbls! with diag_var."""
    L, M, H, H0 = 600, 380, 6, 4
    rng = np.random.default_rng(177)
    Y, A, B = O.toy_matrix(L, M, H, 0.0, rng)
    Y = (B * np.linspace(1.0, 2.5, H)) @ A.T + rng.uniform(0.02, 0.4, (L, 1)) * rng.standard_normal((L, M))
    po = O.vbmf_dual_init(Y, H, H0, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(178), materialize_yhat=False)
    ydt = pkg.VBMF_Y_F32 if mode == "f32" else pkg.VBMF_Y_BF16
    with pkg.capi.Context(L, M, H, y_dtype=ydt) as c:
        c.set_Y(Y)
        Ys = np.ascontiguousarray(c.get_Y())
    po.trYTY = float(np.sum(Ys * Ys))
    pkg.set_defaults(y_dtype=ydt, factor_dtype=pkg.VBMF_FACTOR_AUTO)

    def to_pkg(q):
        p = _to_pkg(pkg, q)
        p.sigmaVecHat, p.etaVec, p.zetaVec = q.sigmaVecHat.copy(), q.etaVec.copy(), q.zetaVec.copy()
        return p
    pg = to_pkg(po)
    d_gpu = pkg.vbmf_dual_(Ys, pg, 8, eps=0.0, diag_var=True, est_priors=True)
    d_ref, n = O.vbmf_dual_(Ys, po, 8, eps=0.0, diag_var=True, est_priors=True)
    tol = 2e-3 if mode == "f32" else 5e-3
    _cmp(f"diag_var run8 {mode}", pg, po, tol, FIELDS + ("sigmaVecHat", "zetaVec"), priors_tol=tol)
    assert pg._last_run[0] == 8 and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6
    pg = to_pkg(po)
    pkg.vbls_(Ys, pg, 4, diag_var=True)
    for _ in range(4):                                                      # examples/mil_util.jl:190-193
        O.dual_updateA(Ys, po, diag_var=True); O.dual_updateCA(po); O.sparse_updateSigma(Ys, po, diag_var=True)
    _cmp(f"diag_var vbls4 {mode}", pg, po, tol, ("ATVecHat", "diagSigmaATVec", "SigmaA", "CA", "beta", "sigmaVecHat", "zetaVec"))


def test_dual_full_cov_as_the_mil_callers_run_it(pkg):
    """examples/mil_util.jl:345-351 runs vbmf_dual! with full_cov = true (M*H <= 3200) and est_priors: the per-column
    H x H blocks on the device against the oracle's dense kron(...) restatement."""
    L, M, H, H0 = 300, 160, 6, 4
    Y, po = _mk(L, M, H, H0, 61)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    pg = _to_pkg(pkg, po)
    d_gpu = pkg.vbmf_dual_(Yf, pg, 10, eps=0.0, full_cov=True, est_priors=True)
    d_ref, n = O.vbmf_dual_(Yf, po, 10, eps=0.0, full_cov=True, est_priors=True)
    _cmp("full_cov run10 f32", pg, po, 2e-3, priors_tol=2e-3)
    assert np.any(pg.SigmaA != np.diag(np.diag(pg.SigmaA))) and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6


def test_dual_full_cov_with_heteroscedastic_rows(pkg):
    """vbmf_dual! with full_cov = true AND diag_var = true (src/vbmf_dual.jl:218-243 with the row-noise forms of :220-222,
    :232-233 -- the sparse model's bodies): 6 sweeps with the prior fits against the oracle's dense restatement.  Unpinned."""
    L, M, H, H0 = 320, 150, 6, 4
    rng = np.random.default_rng(277)
    Y, A, B = O.toy_matrix(L, M, H, 0.0, rng)
    Y = (B * np.linspace(1.0, 2.5, H)) @ A.T + rng.uniform(0.02, 0.4, (L, 1)) * rng.standard_normal((L, M))
    po = O.vbmf_dual_init(Y, H, H0, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(278), materialize_yhat=False)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    pg = _to_pkg(pkg, po)
    pg.sigmaVecHat, pg.etaVec, pg.zetaVec = po.sigmaVecHat.copy(), po.etaVec.copy(), po.zetaVec.copy()
    d_gpu = pkg.vbmf_dual_(Yf, pg, 6, eps=0.0, full_cov=True, diag_var=True, est_priors=True)
    d_ref, n = O.vbmf_dual_(Yf, po, 6, eps=0.0, full_cov=True, diag_var=True, est_priors=True)
    _cmp("full_cov+diag_var run6 f32", pg, po, 2e-3, FIELDS + ("sigmaVecHat", "zetaVec"), priors_tol=2e-3)
    assert np.any(pg.SigmaA != np.diag(np.diag(pg.SigmaA))) and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6


def test_dual_lower_bound_trimmed(pkg):
    """lowerBoundTrimmed of the two-group model (src/vbmf_dual.jl:606-617): only the vec fields are trimmed there, the
    per-group fields stay whole -- so MH, the CA-weighted second moment and H(vec(A')) change, the group terms do not."""
    L, M, H, H0 = 300, 170, 5, 3
    Y, po = _mk(L, M, H, H0, 71)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    O.vbmf_dual_(Yf, po, 8, eps=0.0, est_cb=True, est_priors=True)
    po.ATVecHat = po.ATVecHat.astype(np.float32).astype(np.float64)
    po.AHat = po.ATVecHat.reshape(M, H).copy()
    po.A0Hat, po.A1Hat = po.AHat[:, :H0].copy(), po.AHat[:, H0:].copy()
    for trim in (1e-1, 0.7):
        want, got = O.lowerBoundTrimmed(Yf, po, trim), pkg.lowerBoundTrimmed(Yf, _to_pkg(pkg, po), trim)
        report(f"dual lowerBoundTrimmed trim={trim:g}: gpu {got:.6f} oracle {want:.6f} (untrimmed {O.lowerBound_dual(Yf, po):.6f})")
        # the device's bound carries the same absolute error with and without the mask (fp32 storage of beta / CA): the
        # untrimmed bound's tolerance, 2e-5 of ITS magnitude
        assert abs(got - want) <= 2e-5 * abs(O.lowerBound_dual(Yf, po)) + 1e-3, (trim, got, want)
