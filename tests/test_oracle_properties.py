"""Properties of the CPU oracle that hold independently of any recorded number (CPU-only): they guard the parts of the
restatement the reference's fixtures do not pin (SURVEY.md section 8c "parity unpinned") against plain mistakes --
the fused/faithful orderings agree, the build-defined ELBO ascends, the literal loops of the reference's source agree
with the vectorised restatements, and the diagonal sparse branch is the diagonal of the pinned full-covariance branch."""
import copy

import numpy as np
import pytest

from oracle import vbmf_oracle as O


def _problem(L, M, H, seed, **kw):
    rng = np.random.default_rng(seed)
    Y, A, B = O.toy_matrix(L, M, H, 0.05, rng)
    Y = (B * np.linspace(1.0, 2.5, H)) @ A.T + 0.05 * rng.standard_normal((L, M))
    return Y, O.vbmf_init(Y, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(seed + 1), **kw)


def test_fused_and_faithful_orderings_agree():
    """updateSigma2! literally (Y.^2, 2Y', the M x M product: src/vbmf.jl:153-157) vs the fused form every GPU kernel
    and the CPU baseline's `fused_value` use."""
    Y, p = _problem(60, 45, 4, 1)
    q = O.copy_params(p)
    for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB"):
        setattr(q, f, getattr(p, f).copy())
    O.vbmf_(Y, p, 12, eps=0.0, est_covs=True, est_var=True, fused=False)
    O.vbmf_(Y, q, 12, eps=0.0, est_covs=True, est_var=True, fused=True)
    for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB"):
        assert np.allclose(getattr(p, f), getattr(q, f), rtol=1e-8, atol=1e-12), f
    assert p.sigma2 == pytest.approx(q.sigma2, rel=1e-9)


def test_elbo_ascends_with_frozen_hyperparameters():
    """Coordinate ascent on q(A), q(B) with C_A, C_B, sigma2 fixed cannot lower the bound."""
    Y, p = _problem(80, 50, 3, 2)
    tr = []
    O.vbmf_(Y, p, 15, eps=0.0, est_covs=False, est_var=False, trace=tr)
    e = np.array([t[2] for t in tr])
    assert np.all(np.isfinite(e)) and np.all(np.diff(e) >= -1e-9 * np.abs(e[:-1])), e


def test_delta_is_the_spectral_norm_ratio():
    """delta = norm(old - new) / norm(old) with Julia 0.5's norm(::Matrix) = largest singular value (src/util.jl:27-29)."""
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((30, 4)), rng.standard_normal((30, 4))
    want = np.linalg.svd(b - a, compute_uv=False)[0] / np.linalg.svd(b, compute_uv=False)[0]
    assert O.delta(a, b) == pytest.approx(want, rel=1e-12)
    fro = np.linalg.norm(b - a) / np.linalg.norm(b)
    assert abs(O.delta(a, b) - fro) > 1e-3                     # and it is NOT the Frobenius ratio
    with np.errstate(all="ignore"):
        assert np.isnan(O.delta(np.zeros((3, 2)), np.zeros((3, 2))))   # 0/0 -> NaN -> the loop exits (App. A Q6)


def test_mask_and_termination():
    Y, p = _problem(40, 30, 4, 4, H1=2, labels=[1, 5, 9])
    assert np.all(p.AHat[[1, 5, 9], 2:] == 0.0)               # masked at init (src/vbmf.jl:61)
    _, n, d = O.vbmf_(Y, p, 200, eps=1e-3, est_covs=True, est_var=True)
    assert np.all(p.AHat[[1, 5, 9], 2:] == 0.0) and np.all(p.AHat[[0, 2], 2:] != 0.0)
    assert 1 <= n < 200 and d <= 1e-3                          # while i <= niter && d > eps (:193)
    q = copy.deepcopy(p)
    _, n0, d0 = O.vbmf_(Y, q, 0, eps=1e-3)
    assert n0 == 0 and d0 == 1e-3 + 1.0                        # d = eps + 1 before the loop (:189)


def test_sparse_diagonal_branch_is_the_diagonal_of_the_pinned_full_branch():
    """With the consistent layout (reference_compat=False) and CA a per-column constant, the diagonal branch
    (src/vbmf_sparse.jl:214-239) equals the full-covariance branch (:178-202, pinned by the fixture) whenever the full
    posterior precision is itself diagonal, i.e. B'B + L SigmaB diagonal -- and QS2 is visible as the one difference."""
    rng = np.random.default_rng(5)
    L, M, H = 12, 7, 3
    Q, _ = np.linalg.qr(rng.standard_normal((L, H)))
    B = Q * np.array([1.0, 2.0, 0.5])                           # orthogonal columns: B'B diagonal
    Y = rng.standard_normal((L, M))
    base = O.vbmf_sparse_init(Y, H, ca=0.7, cb=1.0, sigma=1.0, rng=rng, full_cov=True)
    base.BHat = B
    base.SigmaB = np.diag([0.01, 0.02, 0.03])
    full, diag = copy.deepcopy(base), copy.deepcopy(base)
    O.sparse_updateA(Y, full, full_cov=True)
    O.sparse_updateA(Y, diag, full_cov=False, reference_compat=False)
    # sigmaHat = 1 makes QS2 (sigmaHat not multiplying L*SigmaB in the diagonal branch) vanish
    assert np.allclose(full.ATVecHat, diag.ATVecHat, rtol=1e-10) and np.allclose(full.SigmaA, diag.SigmaA, rtol=1e-10)
    base.sigmaHat = 3.0
    full, diag = copy.deepcopy(base), copy.deepcopy(base)
    O.sparse_updateA(Y, full, full_cov=True)
    O.sparse_updateA(Y, diag, full_cov=False, reference_compat=False)
    assert not np.allclose(full.diagSigmaATVec, diag.diagSigmaATVec, rtol=1e-3)       # QS2
    v_full = 3.0 * (np.sum(B * B, axis=0) + L * np.diag(base.SigmaB))
    v_diag = 3.0 * np.sum(B * B, axis=0) + L * np.diag(base.SigmaB)
    assert np.allclose(1.0 / (np.tile(v_full, M) + base.CA), full.diagSigmaATVec)
    assert np.allclose(1.0 / (np.tile(v_diag, M) + base.CA), diag.diagSigmaATVec)


def test_repeat_layout_quirk():
    """repeat(v, inner = M-1) after the first H entries (src/vbmf_sparse.jl:221, App. A QS1)."""
    v = np.array([10.0, 20.0, 30.0])
    got = O.spread_v(v, 4, reference_compat=True)
    assert got.tolist() == [10, 20, 30, 10, 10, 10, 20, 20, 20, 30, 30, 30]
    assert O.spread_v(v, 4, reference_compat=False).tolist() == [10, 20, 30] * 4


def test_heteroscedastic_sigma_matches_the_literal_row_loop():
    """updateSigma!, diag_var = true (src/vbmf_sparse.jl:309-315) written row by row as in the source."""
    rng = np.random.default_rng(6)
    L, M, H = 15, 9, 3
    Y = rng.standard_normal((L, M))
    p = O.vbmf_sparse_init(Y, H, rng=rng, full_cov=False)
    p.SigmaA = np.diag(rng.uniform(0.1, 1.0, H)); p.SigmaB = np.diag(rng.uniform(0.1, 1.0, H))
    q = copy.deepcopy(p)
    O.sparse_updateSigma(Y, p, diag_var=True)
    for l in range(L):
        z = (q.zeta0 + 0.5 * np.sum(Y[l] ** 2) - np.sum(Y[l] * (q.AHat @ q.BHat[l]))
             + 0.5 * np.sum((q.AHat.T @ q.AHat + q.SigmaA) * (np.outer(q.BHat[l], q.BHat[l]) + q.SigmaB)))
        assert p.zetaVec[l] == pytest.approx(z, rel=1e-12)
        assert p.sigmaVecHat[l] == pytest.approx(q.etaVec[l] / z, rel=1e-12)


def test_lower_bound_entropy_term_equals_the_kron_determinant():
    """H(B) = normalEntropy(kron(SigmaB, I_L)) (src/vbmf_sparse.jl:463) restated as L * logdet(SigmaB) (App. A QS5)."""
    rng = np.random.default_rng(7)
    L, H = 4, 3
    X = rng.standard_normal((H, H)); S = X @ X.T + np.eye(H)
    big = np.kron(S, np.eye(L))
    sign, ld = np.linalg.slogdet(big)
    want = 0.5 * (L * H) * (1.0 + O.LN2PI) + 0.5 * ld
    got = O.normalEntropy_matrix_logdet(L * H, L * np.linalg.slogdet(S)[1], clamp=True)
    assert got == pytest.approx(want, rel=1e-12)


def test_scaleY_and_preprocess():
    rng = np.random.default_rng(8)
    Y = rng.standard_normal((9, 40)) * 3.0 + 5.0
    Y[2] = 4.0                                                  # constant row: dropped by preprocess
    s = O.scaleY(Y)
    keep = np.arange(9) != 2
    assert np.allclose(s[keep].mean(axis=1), 0.0, atol=1e-12) and np.allclose(s[keep].var(axis=1, ddof=1), 1.0)
    assert np.all(s[2] == 0.0)
    out, rows = O.preprocess(Y, 0.5, return_rows=True)
    assert rows.tolist() == [0, 1, 3, 4, 5, 6, 7, 8] and np.allclose(out, 0.5 * s[keep])


def test_vbls_keeps_the_basis_fixed():
    Y, p = _problem(50, 35, 3, 9)
    O.vbmf_(Y, p, 10, eps=0.0, est_covs=True, est_var=True)
    Y2 = Y[:, :20].copy()
    q = O.copy_vbmf_params(Y2, p, rng=np.random.default_rng(1))
    assert q.M == 20 and np.array_equal(q.BHat, p.BHat) and q.sigma2 == p.sigma2
    B0 = q.BHat.copy()
    A = O.vbls_(Y2, q, 8)
    assert A is q.AHat and np.array_equal(q.BHat, B0) and np.isfinite(q.sigma2)


@pytest.mark.parametrize("L,M,H", [(9, 6, 1), (9, 1, 3)])
def test_diagonal_branch_equals_full_branch_when_the_precision_is_diagonal_by_shape(L, M, H):
    """with H = 1 the H x H blocks of the full posterior precision are scalars, so the full
    branch IS diagonal; with M = 1 the `repeat(v, inner = M-1)` tail is empty, so the QS1 layout coincides with the
    consistent one.  In the full branch K = sigmaHat (B'B + L SigmaB); the diagonal branch's v differs by QS2 only, which
    vanishes for sigmaHat = 1 -- and for H = 1, M >= 2 the reference layout still equals the consistent tiling."""
    rng = np.random.default_rng(11)
    Y = rng.standard_normal((L, M))
    base = O.vbmf_sparse_init(Y, H, ca=0.6, cb=1.0, sigma=1.0, rng=rng, full_cov=True)
    if H > 1:                                                    # M = 1: make B'B + L SigmaB diagonal so both branches agree
        Q, _ = np.linalg.qr(rng.standard_normal((L, H)))
        base.BHat = Q * np.array([1.0, 2.0, 0.5])
    base.SigmaB = np.diag(rng.uniform(0.01, 0.05, H))
    full, diag, diag_compat = copy.deepcopy(base), copy.deepcopy(base), copy.deepcopy(base)
    O.sparse_updateA(Y, full, full_cov=True)
    O.sparse_updateA(Y, diag, full_cov=False, reference_compat=False)
    O.sparse_updateA(Y, diag_compat, full_cov=False, reference_compat=True)
    for q in (diag, diag_compat):
        assert np.allclose(q.ATVecHat, full.ATVecHat, rtol=1e-10, atol=1e-14)
        assert np.allclose(q.diagSigmaATVec, full.diagSigmaATVec, rtol=1e-10)
        assert np.allclose(q.SigmaA, full.SigmaA, rtol=1e-10)


# ---- grouped ARD variants (src/vbmf_dual.jl, src/vbmf_trial.jl) ---------------------------------
def _toy(L=60, M=40, H=4, seed=1):
    rng = np.random.default_rng(seed)
    Y, _, _ = O.toy_matrix(L, M, H, 0.05, rng)
    return Y


def test_grouped_variants_reduce_to_each_other():
    """vbmf_dual! without prior fits is vbmf_sparse! (same bodies, src/vbmf_dual.jl:216-386 vs src/vbmf_sparse.jl), and
    vbmf_trial! with every row in the first block (M0 = M) is vbmf_dual!, prior fits and lowerBound included."""
    Y = _toy()
    pd = O.vbmf_dual_init(Y, 5, 3, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(3))
    ps = O.vbmf_sparse_init(Y, 5, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(3), full_cov=False)
    O.vbmf_dual_(Y, pd, 10, eps=0.0, est_priors=False)
    O.vbmf_sparse_(Y, ps, 10, eps=0.0)
    assert np.array_equal(pd.AHat, ps.AHat) and np.array_equal(pd.CA, ps.CA) and pd.sigmaHat == ps.sigmaHat
    assert O.lowerBound_dual(Y, pd) == pytest.approx(O.lowerBound(Y, ps), rel=1e-13)
    pt = O.vbmf_trial_init(Y, 5, 3, Y.shape[1], ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(3))
    pd = O.vbmf_dual_init(Y, 5, 3, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(3))
    O.vbmf_trial_(Y, pt, 10, eps=0.0)
    O.vbmf_dual_(Y, pd, 10, eps=0.0)
    assert np.array_equal(pt.AHat, pd.AHat) and (pt.alpha01, pt.beta01, pt.alpha02, pt.beta02) == (pd.alpha00, pd.beta00, pd.alpha01, pd.beta01)
    assert (pt.alpha03, pt.beta03) == (1e-10, 1e-10)                       # the empty block is never fitted
    assert O.lowerBound_trial(Y, pt) == pytest.approx(O.lowerBound_dual(Y, pd), rel=1e-13)


def test_grouped_layout_and_prior_fit_stationarity():
    """The interleaved CA/beta vectors follow the reference's concatenation loops (src/vbmf_trial.jl:178-186), and each
    fitted (alpha0g, beta0g) is a stationary point of the bound: digamma(alpha0g) = log(beta0g_old) + mean E[ln CA_g],
    beta0g = n_g alpha0g / sum(CA_g) (src/vbmf_trial.jl:442-507)."""
    from scipy.special import digamma
    Y = _toy(50, 30, 4, 5)
    M, H, H0, M0 = 30, 6, 2, 11
    p = O.vbmf_trial_init(Y, H, H0, M0, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(9))
    p.CA1, p.CA2, p.CA3 = 1.0 + np.arange(M * H0), 1000.0 + np.arange(M0 * (H - H0)), 5000.0 + np.arange((M - M0) * (H - H0))
    ca = O._trial_join(p.CA1, p.CA2, p.CA3, M, H, H0, M0)
    ref = []                                                               # the reference's loops, verbatim in spirit
    H1 = H - H0
    for m in range(1, M0 + 1):
        ref += list(p.CA1[(m - 1) * H0:m * H0]) + list(p.CA2[(m - 1) * H1:m * H1])
    for m in range(M0 + 1, M + 1):
        ref += list(p.CA1[(m - 1) * H0:m * H0]) + list(p.CA3[(m - 1 - M0) * H1:(m - M0) * H1])
    assert np.array_equal(ca, np.array(ref))
    assert all(np.array_equal(a, b) for a, b in zip(O._trial_split(ca, M, H, H0, M0), (p.CA1, p.CA2, p.CA3)))
    p = O.vbmf_trial_init(Y, H, H0, M0, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(9))
    O.vbmf_trial_(Y, p, 3, eps=0.0, est_priors=True)
    O.trial_updateA(Y, p); O.sparse_updateB(Y, p); O.trial_updateCA(p)
    old = (p.beta01, p.beta02, p.beta03)
    O.trial_updatePriors(p)
    for n, a0, b0, b_old, ap, rates, cag in ((M * H0, p.alpha01, p.beta01, old[0], p.alpha1, p.beta1, p.CA1),
                                            (M0 * H1, p.alpha02, p.beta02, old[1], p.alpha2, p.beta2, p.CA2),
                                            ((M - M0) * H1, p.alpha03, p.beta03, old[2], p.alpha3, p.beta3, p.CA3)):
        assert digamma(a0) == pytest.approx(np.log(b_old) + np.mean(O.gammaELn(ap, rates)), rel=1e-10, abs=1e-10)
        assert b0 == pytest.approx(n * a0 / np.sum(cag), rel=1e-14)


def test_grouped_prior_fit_keeps_the_value_without_a_sign_change():
    """The reference wraps fzero in `try ... end` (src/vbmf_dual.jl:397-400): no root inside [1e-10, 1e10] -> unchanged."""
    rates = np.full(12, 1e-30)                                             # E[ln CA] = digamma(a) + 69: root > 1e10
    assert O._dual_fit_shape(12, np.log(1.0), 0.5, rates, 0.123) == 0.123
    assert O._dual_fit_shape(0, 0.0, 0.5, np.zeros(0), 0.7) == 0.7         # empty group
    x = O._dual_fit_shape(12, np.log(2.0), 1.5, np.full(12, 3.0), 0.123)
    from scipy.special import digamma
    assert digamma(x) == pytest.approx(np.log(2.0) + digamma(1.5) - np.log(3.0), rel=1e-12)


def test_full_covariance_with_row_noise_is_blockwise_and_reduces_to_the_homoscedastic_branch():
    """updateA!, full_cov = true with diag_var = true (src/vbmf_sparse.jl:180-182, 192-193) -- no recorded run pins this branch:
    (i) written with diagm(sigmaVecHat) and kron(eye(M), .) as in the source, it equals M independent H x H solves with
    K = B' diag(sigma) B + L mean(sigma) SigmaB; (ii) with every row precision equal to s it IS the homoscedastic full branch at
    sigmaHat = s (that one is pinned by the reference's recorded sparse run)."""
    rng = np.random.default_rng(16)
    L, M, H = 14, 7, 3
    Y = rng.standard_normal((L, M))
    p = O.vbmf_sparse_init(Y, H, rng=rng)
    p.SigmaB = np.diag(rng.uniform(0.1, 1.0, H)); p.CA = rng.uniform(0.5, 2.0, M * H)
    p.sigmaVecHat = rng.uniform(0.3, 3.0, L)
    q = copy.deepcopy(p)
    O.sparse_updateA(Y, p, full_cov=True, diag_var=True)
    # (i) the source's dense form, literally
    inv_full = np.kron(np.eye(M), q.BHat.T @ np.diag(q.sigmaVecHat) @ q.BHat + L * np.mean(q.sigmaVecHat) * q.SigmaB) + np.diag(q.CA)
    S = np.linalg.inv(inv_full)
    vecA = S @ (q.BHat.T @ np.diag(q.sigmaVecHat) @ Y).T.reshape(M * H)
    assert np.allclose(p.ATVecHat, vecA, rtol=1e-12, atol=1e-14)
    K = q.BHat.T @ (q.sigmaVecHat[:, None] * q.BHat) + L * np.mean(q.sigmaVecHat) * q.SigmaB
    SA = np.zeros((H, H))
    for m in range(M):
        Sm = np.linalg.inv(K + np.diag(q.CA[m * H:(m + 1) * H]))
        SA += Sm
        assert np.allclose(p.diagSigmaATVec[m * H:(m + 1) * H], np.diag(Sm), rtol=1e-11)
        assert np.allclose(p.AHat[m], Sm @ (q.BHat.T @ (q.sigmaVecHat * Y[:, m])), rtol=1e-10, atol=1e-13)
    assert np.allclose(p.SigmaA, SA, rtol=1e-11)
    # (ii) constant row precision = the homoscedastic branch
    s = 1.7
    a, b = copy.deepcopy(q), copy.deepcopy(q)
    a.sigmaVecHat = np.full(L, s); b.sigmaHat = s
    O.sparse_updateA(Y, a, full_cov=True, diag_var=True)
    O.sparse_updateA(Y, b, full_cov=True, diag_var=False)
    for f in ("ATVecHat", "diagSigmaATVec", "SigmaA"):
        assert np.allclose(getattr(a, f), getattr(b, f), rtol=1e-11, atol=1e-14), f
