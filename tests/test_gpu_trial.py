"""GPU parity of the three-group ARD variant (src/vbmf_trial.jl, full_cov=false, diag_var=false)
against the oracle.  PARITY UNPINNED, as for the two-group model (tests/test_gpu_dual.py): no recorded vbmf_trial! run
exists and the hyper-prior fits call the unpinned Roots.jl."""
import numpy as np
import pytest

import __graft_entry__ as G
from oracle import vbmf_oracle as O
from tests.helpers import relF, report

pytestmark = pytest.mark.gpu
PRI = ("alpha01", "beta01", "alpha02", "beta02", "alpha03", "beta03")


@pytest.fixture(scope="module")
def pkg():
    G.build()
    return G.load_package()


def _mk(L, M, H, H0, M0, seed):
    rng = np.random.default_rng(seed)
    Y, A, B = O.toy_matrix(L, M, H, 0.05, rng)
    Y = (B * np.linspace(1.0, 2.5, H)) @ A.T + 0.05 * rng.standard_normal((L, M))
    po = O.vbmf_trial_init(Y, H, H0, M0, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(seed + 1), materialize_yhat=False)
    return Y, po


SCAL = ("L", "M", "M0", "M1", "H", "MH", "H0", "H1") + PRI + ("alpha1", "alpha2", "alpha3", "gamma0", "delta0", "gamma",
                                                               "sigmaHat", "eta0", "zeta0", "eta", "zeta", "trYTY")
ARRS = ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "A1Hat", "A2Hat", "A3Hat", "BHat", "SigmaB", "CA", "beta", "CA1",
        "CA2", "CA3", "beta1", "beta2", "beta3", "CB", "delta")
FIELDS = ("ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta")
GROUPS = ("CA", "beta", "CA1", "CA2", "CA3", "beta1", "beta2", "beta3")


def _to_pkg(pkg, po):
    p = pkg.vbmf_trial_parameters()
    for f in SCAL:
        setattr(p, f, getattr(po, f))
    for f in ARRS:
        setattr(p, f, np.array(getattr(po, f), copy=True))
    p.alpha = np.array([po.alpha1, po.alpha2, po.alpha3])
    return p


def _cmp(tag, pg, po, tol, fields=FIELDS, priors_tol=None):
    errs = {f: relF(getattr(pg, f), getattr(po, f)) for f in fields if np.size(getattr(po, f))}
    errs["sigmaHat"] = abs(pg.sigmaHat - po.sigmaHat) / po.sigmaHat
    if priors_tol is not None:
        for f in PRI:
            errs[f] = abs(getattr(pg, f) - getattr(po, f)) / abs(getattr(po, f))
    report(f"trial {tag}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    bad = {k: v for k, v in errs.items() if not v <= (priors_tol if k in PRI else tol)}
    assert not bad, (tag, bad)


@pytest.mark.parametrize("L,M,H,H0,M0", [(12, 20, 3, 1, 7), (300, 170, 5, 3, 60), (500, 260, 40, 16, 200)])
def test_trial_each_update_f32(pkg, L, M, H, H0, M0):
    Y, po = _mk(L, M, H, H0, M0, 240 + H)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    tag = f"{L}x{M} H{H}/{H0} M0={M0}"
    for sweep in range(3):
        pg = _to_pkg(pkg, po)
        pkg.trial_updateA_(Yf, pg); O.trial_updateA(Yf, po)
        _cmp(f"{tag} s{sweep} updateA", pg, po, 5e-5, ("ATVecHat", "diagSigmaATVec", "SigmaA", "A1Hat", "A2Hat", "A3Hat"))
        pg = _to_pkg(pkg, po)
        pkg.trial_updateB_(Yf, pg); O.sparse_updateB(Yf, po)
        _cmp(f"{tag} s{sweep} updateB", pg, po, 5e-5, ("BHat", "SigmaB"))
        pg = _to_pkg(pkg, po)
        pkg.trial_updateCA_and_priors_(pg, Y=Yf); O.trial_updateCA(po); O.trial_updatePriors(po)
        _cmp(f"{tag} s{sweep} updateCA+priors", pg, po, 5e-5, GROUPS, priors_tol=1e-5)
        assert (pg.alpha1, pg.alpha2, pg.alpha3) == (po.alpha1, po.alpha2, po.alpha3)     # what this updateCA! used (:359-361)
        pg = _to_pkg(pkg, po)
        pkg.trial_updateCB_(pg, Y=Yf); O.sparse_updateCB(po)
        _cmp(f"{tag} s{sweep} updateCB", pg, po, 5e-5, ("CB", "delta"))
        pg = _to_pkg(pkg, po)
        pkg.trial_updateSigma_(Yf, pg); O.sparse_updateSigma(Yf, po)
        _cmp(f"{tag} s{sweep} updateSigma", pg, po, 5e-4, ())
        assert abs(pg.zeta - po.zeta) / po.zeta < 5e-4


def test_trial_groups_use_their_own_priors(pkg):
    L, M, H, H0, M0 = 120, 70, 6, 2, 25
    Y, po = _mk(L, M, H, H0, M0, 5)
    po.alpha01, po.beta01, po.alpha02, po.beta02, po.alpha03, po.beta03 = 0.7, 0.02, 3.5, 1.25, 1.5, 0.5
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg(pkg, po)
    pkg.trial_updateCA_(pg, Y=Y); O.trial_updateCA(po)
    _cmp("three priors updateCA", pg, po, 1e-6, GROUPS)
    assert (pg.alpha1, pg.alpha2, pg.alpha3) == (1.2, 4.0, 2.0)
    assert tuple(getattr(pg, k) for k in PRI) == (0.7, 0.02, 3.5, 1.25, 1.5, 0.5)          # no fit without est_priors


@pytest.mark.parametrize("mode,est_priors", [("f32", True), ("bf16x2", True), ("f32", False)])
def test_trial_run_and_lower_bound(pkg, mode, est_priors):
    L, M, H, H0, M0 = 600, 380, 6, 4, 150
    Y, po = _mk(L, M, H, H0, M0, 221)
    ydt = pkg.VBMF_Y_F32 if mode == "f32" else pkg.VBMF_Y_BF16
    with pkg.capi.Context(L, M, H, y_dtype=ydt) as c:
        c.set_Y(Y)
        Ys = np.ascontiguousarray(c.get_Y())
    po.trYTY = float(np.sum(Ys * Ys))
    pkg.set_defaults(y_dtype=ydt, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg(pkg, po)
    d_gpu = pkg.vbmf_trial_(Ys, pg, 15, eps=0.0, est_cb=True, est_priors=est_priors)
    d_ref, n = O.vbmf_trial_(Ys, po, 15, eps=0.0, est_cb=True, est_priors=est_priors)
    tol = 2e-3 if mode == "f32" else 5e-3
    _cmp(f"run15 {mode} est_priors={est_priors}", pg, po, tol, priors_tol=tol)
    assert pg._last_run[0] == 15 and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6
    if est_priors:
        assert len({po.alpha01, po.alpha02, po.alpha03, 1e-10}) == 4                      # three different fits
    else:
        assert tuple(getattr(pg, k) for k in PRI) == (1e-10,) * 6
    lb_gpu = pkg.lowerBound_trial(Ys, pg)
    lb_ref = O.lowerBound_trial(Ys, po)
    report(f"trial lowerBound {mode} est_priors={est_priors}: gpu {lb_gpu:.6f} oracle {lb_ref:.6f}")
    assert abs(lb_gpu - lb_ref) <= 2e-3 * abs(lb_ref)
    lb2 = pkg.lowerBound_trial(Ys, _to_pkg(pkg, po))         # the ORACLE's state on the device: the bound itself
    assert abs(lb2 - lb_ref) <= 2e-5 * abs(lb_ref), (lb2, lb_ref)


def test_trial_with_all_rows_in_one_block_is_the_two_group_model(pkg):
    """M0 = M: A3 is empty and vbmf_trial! is vbmf_dual! (bit for bit on the device, prior fits included)."""
    L, M, H, H0 = 400, 250, 8, 3
    Y, pt = _mk(L, M, H, H0, M, 77)
    pd = O.vbmf_dual_init(Y, H, H0, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(78), materialize_yhat=False)
    assert np.array_equal(pd.AHat, pt.AHat) and np.array_equal(pd.BHat, pt.BHat)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_BF16, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    gt = _to_pkg(pkg, pt)
    pkg.vbmf_trial_(Y, gt, 8, eps=0.0, est_priors=True)
    from tests.test_gpu_dual import _to_pkg as dual_to_pkg
    gd = dual_to_pkg(pkg, pd)
    pkg.vbmf_dual_(Y, gd, 8, eps=0.0, est_priors=True)
    for f in FIELDS:
        assert np.array_equal(getattr(gt, f), getattr(gd, f)), f
    assert (gt.alpha01, gt.beta01, gt.alpha02, gt.beta02) == (gd.alpha00, gd.beta00, gd.alpha01, gd.beta01)
    assert gt.sigmaHat == gd.sigmaHat and (gt.alpha03, gt.beta03) == (1e-10, 1e-10)       # the empty group is never fitted


def test_trial_fixed_basis_edges_and_errors(pkg):
    """vbls! on the three-group model (examples/mil_util.jl:194-197), degenerate row splits, argument errors."""
    L, M, H, H0 = 260, 150, 5, 2
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    for M0 in (40, 0, M):
        Y, po = _mk(L, M, H, H0, M0, 31 + M0)
        Yf = Y.astype(np.float32).astype(np.float64)
        po.trYTY = float(np.sum(Yf * Yf))
        O.vbmf_trial_(Yf, po, 3, eps=0.0)
        pg = _to_pkg(pkg, po)
        A = pkg.vbls_(Yf, pg, 6)
        O.vbls_trial_(Yf, po, 6)
        _cmp(f"vbls M0={M0}", pg, po, 2e-4, ("ATVecHat", "diagSigmaATVec", "SigmaA", "CA", "beta"))
        assert A is pg.AHat
        pg = _to_pkg(pkg, po)
        pkg.vbmf_trial_(Yf, pg, 4, eps=0.0, est_priors=True)
        O.vbmf_trial_(Yf, po, 4, eps=0.0, est_priors=True)
        _cmp(f"run4 M0={M0}", pg, po, 1e-3, priors_tol=1e-3)
    hyper = dict(alpha0=1e-10, beta0=1e-10, gamma0=1e-10, delta0=1e-10, eta0=1e-10, zeta0=1e-10)
    pri = {k: 1.0 for k in pkg.capi.Context.TRIAL_KEYS}
    with pkg.capi.Context(L, M, H, y_dtype=pkg.VBMF_Y_F32, variant=pkg.capi.VBMF_VARIANT_TRIAL_DIAG) as c:
        c.set_Y(Y)
        with pytest.raises(pkg.VbmfError):
            c.trial_set_priors(2, 10, pri)                                 # no state yet
        c.sparse_set_state(po.ATVecHat, po.diagSigmaATVec, po.CA, po.beta, po.BHat, po.SigmaB, po.CB, po.delta, po.sigmaHat,
                           po.zeta, hyper)
        with pytest.raises(pkg.VbmfError, match="M0"):
            c.trial_set_priors(2, M + 1, pri)
        with pytest.raises(pkg.VbmfError, match="H0"):
            c.trial_set_priors(H + 1, 10, pri)
        with pytest.raises(pkg.VbmfError):
            c.trial_set_priors(2, 10, dict(pri, beta03=0.0))
        with pytest.raises(pkg.VbmfError):
            c.dual_run(1)                                                  # a three-group context is not a two-group one
        c.trial_set_priors(2, 10, pri)
        assert c.trial_get_priors()[:2] == (2, 10)


def test_trial_heteroscedastic_run(pkg):
    """vbmf_trial! with diag_var = true (the sparse model's heteroscedastic bodies, src/vbmf_trial.jl:279-283, 297-298,
    329-334, 420-427) and the prior fits, 8 sweeps."""
    L, M, H, H0, M0 = 600, 380, 6, 4, 150
    rng = np.random.default_rng(277)
    Y, A, B = O.toy_matrix(L, M, H, 0.0, rng)
    Y = (B * np.linspace(1.0, 2.5, H)) @ A.T + rng.uniform(0.02, 0.4, (L, 1)) * rng.standard_normal((L, M))
    po = O.vbmf_trial_init(Y, H, H0, M0, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(278), materialize_yhat=False)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg(pkg, po)
    pg.sigmaVecHat, pg.etaVec, pg.zetaVec = po.sigmaVecHat.copy(), po.etaVec.copy(), po.zetaVec.copy()
    d_gpu = pkg.vbmf_trial_(Yf, pg, 8, eps=0.0, diag_var=True, est_priors=True)
    d_ref, n = O.vbmf_trial_(Yf, po, 8, eps=0.0, diag_var=True, est_priors=True)
    _cmp("diag_var run8 f32", pg, po, 2e-3, FIELDS + ("sigmaVecHat", "zetaVec"), priors_tol=2e-3)
    assert pg._last_run[0] == 8 and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6
    assert np.corrcoef(np.log(pg.sigmaVecHat), np.log(po.sigmaVecHat))[0, 1] > 0.999


def test_trial_full_cov_run(pkg):
    """vbmf_trial! with full_cov = true (src/vbmf_trial.jl:252-277) and the prior fits."""
    L, M, H, H0, M0 = 300, 160, 6, 4, 70
    Y, po = _mk(L, M, H, H0, M0, 63)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    pg = _to_pkg(pkg, po)
    d_gpu = pkg.vbmf_trial_(Yf, pg, 10, eps=0.0, full_cov=True, est_priors=True)
    d_ref, n = O.vbmf_trial_(Yf, po, 10, eps=0.0, full_cov=True, est_priors=True)
    _cmp("full_cov run10 f32", pg, po, 2e-3, priors_tol=2e-3)
    assert np.any(pg.SigmaA != np.diag(np.diag(pg.SigmaA))) and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6


@pytest.mark.parametrize("L,M,H,H0,M0", [(37, 41, 1, 0, 13), (50, 33, 1, 1, 5), (65, 97, 17, 9, 96), (130, 70, 33, 32, 1),
                                         (90, 129, 64, 31, 64)])
def test_trial_odd_shapes_all_branches(pkg, L, M, H, H0, M0):
    """Shapes off every tile boundary (H = 1, 17, 33, 64; M, M0 not multiples of 32; a one-row block): the grouped CA update
    with prior fits, then updateA! through the diagonal AND the per-column full-covariance branch, against the oracle."""
    Y, po = _mk(L, M, H, H0, M0, 900 + H)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    tag = f"odd {L}x{M} H{H}/{H0} M0={M0}"
    O.vbmf_trial_(Yf, po, 2, eps=0.0, est_priors=True)
    pg = _to_pkg(pkg, po)
    pkg.trial_updateCA_and_priors_(pg, Y=Yf); O.trial_updateCA(po); O.trial_updatePriors(po)
    _cmp(f"{tag} updateCA+priors", pg, po, 5e-5, GROUPS, priors_tol=1e-5)
    for full in (False, True):
        qo = O.vbmf_trial_init(Yf, H, H0, M0, rng=np.random.default_rng(1), materialize_yhat=False)
        for f in SCAL + ARRS:
            v = getattr(po, f)
            setattr(qo, f, v.copy() if isinstance(v, np.ndarray) else v)
        pg = _to_pkg(pkg, qo)
        pkg.trial_updateA_(Yf, pg, full_cov=full); O.trial_updateA(Yf, qo, full_cov=full)
        _cmp(f"{tag} updateA full_cov={full}", pg, qo, 5e-5, ("ATVecHat", "diagSigmaATVec", "SigmaA", "A1Hat", "A2Hat", "A3Hat"))
        pg = _to_pkg(pkg, qo)
        pkg.trial_updateB_(Yf, pg); O.sparse_updateB(Yf, qo)
        _cmp(f"{tag} updateB after full_cov={full}", pg, qo, 5e-5, ("BHat", "SigmaB"))


def test_trial_lower_bound_trimmed(pkg):
    """lowerBoundTrimmed of the three-group model (src/vbmf_trial.jl:687-698), as test_dual_lower_bound_trimmed."""
    L, M, H, H0, M0 = 300, 170, 5, 3, 60
    Y, po = _mk(L, M, H, H0, M0, 81)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    O.vbmf_trial_(Yf, po, 8, eps=0.0, est_cb=True, est_priors=True)
    po.ATVecHat = po.ATVecHat.astype(np.float32).astype(np.float64)
    po.AHat = po.ATVecHat.reshape(M, H).copy()
    po.A1Hat, po.A2Hat, po.A3Hat = po.AHat[:, :H0].copy(), po.AHat[:M0, H0:].copy(), po.AHat[M0:, H0:].copy()
    for trim in (1e-1, 0.7):
        want, got = O.lowerBoundTrimmed(Yf, po, trim), pkg.lowerBoundTrimmed(Yf, _to_pkg(pkg, po), trim)
        report(f"trial lowerBoundTrimmed trim={trim:g}: gpu {got:.6f} oracle {want:.6f}")
        # the device's bound carries the same absolute error with and without the mask (fp32 storage of beta / CA): the
        # untrimmed bound's tolerance, 2e-5 of ITS magnitude
        assert abs(got - want) <= 2e-5 * abs(O.lowerBound_trial(Yf, po)) + 1e-3, (trim, got, want)
