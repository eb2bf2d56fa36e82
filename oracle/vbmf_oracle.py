"""CPU oracle: fp64 NumPy restatement of the reference's VB matrix-factorization path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; the product path (vbmatrixfactorization.jl_amd) never does and
fails loudly when the HIP library is missing.

Parity status: PINNED for the basic path (updateA!/updateB!/updateCA!/updateCB!/updateSigma2! and the
est_covs=est_var=true driver) and for the sparse full_cov=true path by the reference's own recorded
trajectories (tests/golden/*.npz, extracted from examples/data/{vbmf_test,sparse_test}/*.jld);
see tests/test_oracle_golden.py.  UNPINNED (no reference number exists anywhere): the convergence
scalar d, label/H1 masking, est_covs/est_var=false, the sparse diagonal branch, lowerBound, and the
basic-model ELBO (the reference defines none).

The reference is Julia 0.5 source (cannot run in this pipeline: no julia binary).  Every function
cites the reference lines it follows; paths are relative to /root/reference.

Conventions: arrays are NumPy float64, Y is (L, M), AHat (M, H), BHat (L, H).  `labels` are 0-based
row indices of AHat here (the reference's are 1-based Julia indices).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, fields
from typing import Optional

import numpy as np

LN2PI = math.log(2.0 * math.pi)            # src/util.jl:1
EPS0 = 4.9406564584124654e-324             # Julia eps(0.0), src/util.jl:120-122


# ----------------------------------------------------------------------------------------------
# numeric helpers  (src/util.jl)
# ----------------------------------------------------------------------------------------------
def norm2(x):
    """src/util.jl:8-19 -- sum(x.^2)."""
    return float(np.sum(np.square(x)))


def spectral_norm(X):
    """Julia 0.5 `norm(::Matrix)` = induced 2-norm (largest singular value); vectors: 2-norm."""
    X = np.asarray(X)
    if X.ndim == 1:
        return float(np.sqrt(np.sum(X * X)))
    return float(np.linalg.norm(X, 2))


def delta(new, old):
    """src/util.jl:27-29 -- norm(old-new)/norm(old) with operator 2-norms."""
    with np.errstate(divide="ignore", invalid="ignore"):          # Julia: 0/0 = NaN, x/0 = Inf (no exception)
        return float(np.float64(spectral_norm(old - new)) / np.float64(spectral_norm(old)))


def traceXTY(X, Y):
    """src/util.jl:104-106."""
    return float(np.sum(X * Y))


def normalEntropy_matrix_logdet(m, logdet, clamp=True):
    """src/util.jl:113-125 given log(det Sigma); det is clamped to eps(0.0) from below."""
    if clamp:
        logdet = max(logdet, math.log(EPS0))
    return m / 2 + m / 2 * LN2PI + 0.5 * logdet


def normalEntropy_diag(diagSigma):
    """src/util.jl:133-137."""
    n = diagSigma.shape[0]
    return n / 2 + n / 2 * LN2PI + 0.5 * float(np.sum(np.log(diagSigma)))


def gammaEntropy(a, b):
    """src/util.jl:144-146 (verbatim, including the +log(b) sign)."""
    from scipy.special import digamma, gammaln
    return a + np.log(b) + gammaln(a) + (1 - a) * digamma(a)


def gammaELn(a, b):
    """src/util.jl:153-155."""
    from scipy.special import digamma
    return digamma(a) - np.log(b)


# ----------------------------------------------------------------------------------------------
# basic VBMF  (src/vbmf.jl)
# ----------------------------------------------------------------------------------------------
@dataclass
class vbmf_parameters:
    """src/vbmf.jl:22-40 -- same field names and order."""
    L: int = 0
    M: int = 0
    H: int = 0
    H1: int = 0
    labels: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))
    AHat: Optional[np.ndarray] = None
    BHat: Optional[np.ndarray] = None
    SigmaA: Optional[np.ndarray] = None
    SigmaB: Optional[np.ndarray] = None
    CA: Optional[np.ndarray] = None
    CB: Optional[np.ndarray] = None
    invCA: Optional[np.ndarray] = None
    invCB: Optional[np.ndarray] = None
    sigma2: float = 1.0
    YHat: Optional[np.ndarray] = None


def copy_params(p):
    """src/vbmf.jl:80-88 -- SHALLOW copy: arrays are shared."""
    q = type(p)()
    for f in fields(p):
        setattr(q, f.name, getattr(p, f.name))
    return q


def _mask(AHat, labels, H1):
    """src/vbmf.jl:61,101 -- AHat[labels, end-H1+1:end] = 0."""
    if H1 > 0 and len(labels) > 0:
        AHat[np.asarray(labels, dtype=np.int64), AHat.shape[1] - H1:] = 0.0


def vbmf_init(Y, H, ca=1.0, cb=1.0, sigma2=1.0, H1=0, labels=(), rng=None, materialize_yhat=True):
    """src/vbmf.jl:48-73.  `rng` (np.random.Generator) replaces Julia's global MersenneTwister."""
    rng = np.random.default_rng(0) if rng is None else rng
    p = vbmf_parameters()
    L, M = Y.shape
    p.L, p.M, p.H, p.H1 = L, M, H, H1
    p.labels = np.asarray(labels, dtype=np.int64)
    p.AHat = rng.standard_normal((M, H))
    _mask(p.AHat, p.labels, H1)
    p.BHat = rng.standard_normal((L, H))
    p.SigmaA = np.zeros((H, H))
    p.SigmaB = np.zeros((H, H))
    p.CA = ca * np.eye(H)
    p.CB = cb * np.eye(H)
    p.invCA = np.linalg.inv(p.CA)
    p.invCB = np.linalg.inv(p.CB)
    p.sigma2 = float(sigma2)
    p.YHat = p.BHat @ p.AHat.T if materialize_yhat else None
    return p


def updateA(Y, p):
    """src/vbmf.jl:95-102."""
    p.SigmaA = p.sigma2 * np.linalg.inv(p.BHat.T @ p.BHat + p.L * p.SigmaB + p.sigma2 * p.invCA)
    p.AHat = ((Y.T @ p.BHat) @ p.SigmaA) / p.sigma2
    _mask(p.AHat, p.labels, p.H1)


def updateB(Y, p):
    """src/vbmf.jl:109-113."""
    p.SigmaB = p.sigma2 * np.linalg.inv(p.AHat.T @ p.AHat + p.M * p.SigmaA + p.sigma2 * p.invCB)
    p.BHat = ((Y @ p.AHat) @ p.SigmaB) / p.sigma2


def updateYHat(p):
    """src/vbmf.jl:120-122."""
    p.YHat = p.BHat @ p.AHat.T


def updateCA(p):
    """src/vbmf.jl:129-134 -- CA diagonal written in place (Q2), invCA rebound."""
    for h in range(p.H):
        p.CA[h, h] = norm2(p.AHat[:, h]) / p.M + p.SigmaA[h, h]
    p.invCA = np.linalg.inv(p.CA)


def updateCB(p):
    """src/vbmf.jl:141-146."""
    for h in range(p.H):
        p.CB[h, h] = norm2(p.BHat[:, h]) / p.L + p.SigmaB[h, h]
    p.invCB = np.linalg.inv(p.CB)


def updateSigma2(Y, p):
    """src/vbmf.jl:153-157 -- faithful: Y.^2 temporary, 2*Y' temporary, the M x M product."""
    p.sigma2 = (norm2(Y) - np.trace(((2 * Y.T) @ p.BHat) @ p.AHat.T)
                + np.trace((p.AHat.T @ p.AHat + p.M * p.SigmaA) @ (p.BHat.T @ p.BHat + p.L * p.SigmaB))
                ) / (p.L * p.M)


def updateSigma2_fused(Y, p, trYY, Q):
    """Same value as updateSigma2 with tr(Y'BA') = sum((Y A) .* B) and a cached ||Y||^2."""
    p.sigma2 = (trYY - 2.0 * float(np.sum(Q * p.BHat))
                + traceXTY(p.AHat.T @ p.AHat + p.M * p.SigmaA, p.BHat.T @ p.BHat + p.L * p.SigmaB)
                ) / (p.L * p.M)


def elbo_basic(Y, p, trYY=None):
    """Build-defined ELBO of the basic model (the reference has none: SURVEY section 8 row A10).

    Gaussian likelihood/priors/posteriors of src/vbmf.jl:166-170; point estimates CA, CB, sigma2.
    PARITY UNPINNED.  Undefined (-inf) before the first sweep (SigmaA = SigmaB = 0).
    """
    L, M, H = p.L, p.M, p.H
    trYY = norm2(Y) if trYY is None else trYY
    GA = p.AHat.T @ p.AHat + M * p.SigmaA
    GB = p.BHat.T @ p.BHat + L * p.SigmaB
    resid = trYY - 2.0 * float(np.sum((Y @ p.AHat) * p.BHat)) + traceXTY(GA, GB)
    ca, cb = np.diag(p.CA), np.diag(p.CB)
    sA, ldA = np.linalg.slogdet(p.SigmaA)
    sB, ldB = np.linalg.slogdet(p.SigmaB)
    if sA <= 0 or sB <= 0:
        return -math.inf
    F = -(L * M / 2) * math.log(2 * math.pi * p.sigma2) - resid / (2 * p.sigma2)
    F += -(M / 2) * float(np.sum(np.log(ca))) - 0.5 * float(np.sum(np.diag(GA) / ca)) + (M / 2) * ldA + M * H / 2
    F += -(L / 2) * float(np.sum(np.log(cb))) - 0.5 * float(np.sum(np.diag(GB) / cb)) + (L / 2) * ldB + L * H / 2
    return F


def vbmf_(Y, p, niter, eps=1e-6, est_covs=False, est_var=False, fused=False, trace=None):
    """vbmf! -- src/vbmf.jl:175-231 (logging omitted).  Returns (p, iterations_done, d).

    fused=False follows the reference's operation order and temporaries ("faithful", the timed CPU
    baseline); fused=True is the algebraically identical two-pass form the GPU computes.
    `trace`, if a list, receives (d, sigma2, elbo) per sweep.
    """
    old = p.BHat                              # :187-188
    d = eps + 1.0                             # :189
    i = 1
    trYY = norm2(Y) if (fused or trace is not None) else None
    while i <= niter and d > eps:             # :193
        if fused:
            GB = p.BHat.T @ p.BHat
            p.SigmaA = p.sigma2 * np.linalg.inv(GB + p.L * p.SigmaB + p.sigma2 * p.invCA)
            p.AHat = ((Y.T @ p.BHat) @ p.SigmaA) / p.sigma2
            _mask(p.AHat, p.labels, p.H1)
            GA = p.AHat.T @ p.AHat
            p.SigmaB = p.sigma2 * np.linalg.inv(GA + p.M * p.SigmaA + p.sigma2 * p.invCB)
            Q = Y @ p.AHat
            p.BHat = (Q @ p.SigmaB) / p.sigma2
        else:
            updateA(Y, p)
            updateB(Y, p)
        if est_covs:
            updateCA(p)
            updateCB(p)
        if est_var:
            if fused:
                updateSigma2_fused(Y, p, trYY, Q)
            else:
                updateSigma2(Y, p)
        d = delta(p.BHat, old)                # :211
        old = p.BHat                          # :212
        if trace is not None:
            trace.append((d, p.sigma2, elbo_basic(Y, p, trYY)))
        i += 1
    return p, i - 1, d


def vbmf(Y, p_in, niter, **kw):
    """src/vbmf.jl:238-248."""
    p = copy_params(p_in)
    # the reference's shallow copy shares CA/CB whose diagonals are then written in place (Q2);
    # copy those two so oracle callers can reuse p_in -- results are unaffected.
    p.CA, p.CB = p.CA.copy(), p.CB.copy()
    return vbmf_(Y, p, niter, **kw)


def vbls_(Y, p, niter):
    """examples/mil_util.jl:179-203, vbmf_parameters branch: A/CA/sigma2 sweeps with B frozen."""
    for _ in range(niter):
        updateA(Y, p)
        updateCA(p)
        updateSigma2(Y, p)
    return p.AHat


def scaleY(Y):
    """src/util.jl:36-54: rows standardised with the row mean and the (n-1) row variance; variances <= 1e-15 -> 1,
    centred entries <= 1e-8 in magnitude -> 0."""
    Y = np.asarray(Y, dtype=np.float64)
    mu = Y.mean(axis=1, keepdims=True)
    den = Y.var(axis=1, ddof=1, keepdims=True)
    den = np.where(np.abs(den) <= 1e-15, 1.0, den)
    nom = Y - mu
    nom[np.abs(nom) <= 1e-8] = 0.0
    return nom / np.sqrt(den)


def preprocess(Y, lam, return_rows=False):
    """src/util.jl:73-86: scaleY, drop the rows whose absolute sum is < 1e-5, multiply by lambda."""
    sY = scaleY(Y)
    used = np.nonzero(np.abs(sY).sum(axis=1) >= 1e-5)[0]
    out = lam * sY[used, :]
    return (out, used) if return_rows else out


def copy_vbmf_params(Y, old, rng=None):
    """examples/mil_util.jl:212-236: fresh parameters for a new Y keeping BHat, SigmaB, CB, invCB [, gamma, delta]."""
    if isinstance(old, vbmf_parameters):
        p = vbmf_init(Y, old.H, sigma2=old.sigma2, rng=rng, materialize_yhat=False)
        p.BHat, p.SigmaB = old.BHat.copy(), old.SigmaB.copy()
        p.CB, p.invCB = old.CB.copy(), old.invCB.copy()
        return p
    p = vbmf_sparse_init(Y, old.H, alpha0=old.alpha0, beta0=old.beta0, gamma0=old.gamma0, delta0=old.delta0,
                         eta0=old.eta0, zeta0=old.zeta0, rng=rng, full_cov=False, materialize_yhat=False)
    p.BHat, p.SigmaB, p.CB = old.BHat.copy(), old.SigmaB.copy(), old.CB.copy()
    p.gamma, p.delta = old.gamma, old.delta.copy()
    return p


def vbls_sparse_(Y, p, niter, reference_compat=True, diag_var=False):
    """examples/mil_util.jl:187-190, vbmf_sparse_parameters branch (full_cov=false)."""
    for _ in range(niter):
        sparse_updateA(Y, p, full_cov=False, reference_compat=reference_compat, diag_var=diag_var)
        sparse_updateCA(p)
        sparse_updateSigma(Y, p, diag_var=diag_var)
    return p.AHat


# ----------------------------------------------------------------------------------------------
# ARD-sparse VBMF  (src/vbmf_sparse.jl), diag_var=false only
# ----------------------------------------------------------------------------------------------
@dataclass
class vbmf_sparse_parameters:
    """src/vbmf_sparse.jl:47-90 (heteroscedastic fields kept for layout; unused: diag_var=false)."""
    L: int = 0
    M: int = 0
    H: int = 0
    MH: int = 0
    H1: int = 0
    labels: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))
    AHat: Optional[np.ndarray] = None
    ATVecHat: Optional[np.ndarray] = None
    SigmaATVec: Optional[np.ndarray] = None
    diagSigmaATVec: Optional[np.ndarray] = None
    invSigmaATVec: Optional[np.ndarray] = None
    SigmaA: Optional[np.ndarray] = None
    BHat: Optional[np.ndarray] = None
    SigmaB: Optional[np.ndarray] = None
    CA: Optional[np.ndarray] = None
    alpha0: float = 1e-10
    beta0: float = 1e-10
    alpha: float = 0.0
    beta: Optional[np.ndarray] = None
    CB: Optional[np.ndarray] = None
    gamma0: float = 1e-10
    delta0: float = 1e-10
    gamma: float = 0.0
    delta: Optional[np.ndarray] = None
    sigmaHat: float = 1.0
    eta0: float = 1e-10
    zeta0: float = 1e-10
    eta: float = 0.0
    zeta: float = 0.0
    sigmaVecHat: Optional[np.ndarray] = None
    etaVec: Optional[np.ndarray] = None
    zetaVec: Optional[np.ndarray] = None
    YHat: Optional[np.ndarray] = None
    trYTY: float = 0.0


def vbmf_sparse_init(Y, H, ca=1.0, alpha0=1e-10, beta0=1e-10, cb=1.0, gamma0=1e-10, delta0=1e-10,
                     sigma=1.0, eta0=1e-10, zeta0=1e-10, H1=0, labels=(), rng=None, full_cov=True,
                     materialize_yhat=True):
    """src/vbmf_sparse.jl:101-153.  full_cov=False skips the two eye(MH) allocations (QS8)."""
    rng = np.random.default_rng(0) if rng is None else rng
    p = vbmf_sparse_parameters()
    L, M = Y.shape
    p.L, p.M, p.H, p.MH, p.H1 = L, M, H, M * H, H1
    p.labels = np.asarray(labels, dtype=np.int64)
    p.AHat = rng.standard_normal((M, H))
    _mask(p.AHat, p.labels, H1)
    p.ATVecHat = p.AHat.reshape(M * H).copy()          # vec(A'), index m*H+h  (:119)
    p.SigmaATVec = np.eye(M * H) if full_cov else None
    p.diagSigmaATVec = np.ones(M * H)
    p.invSigmaATVec = np.eye(M * H) if full_cov else None
    p.SigmaA = np.zeros((H, H))
    p.BHat = rng.standard_normal((L, H))
    p.SigmaB = np.zeros((H, H))
    p.CA = ca * np.ones(M * H)
    p.alpha0, p.beta0 = alpha0, beta0
    p.alpha = alpha0 + 0.5
    p.beta = beta0 * np.ones(M * H)
    p.CB = cb * np.ones(H)
    p.gamma0, p.delta0 = gamma0, delta0
    p.gamma = gamma0 + L / 2
    p.delta = delta0 * np.ones(H)
    p.sigmaHat = float(sigma)
    p.eta0, p.zeta0 = eta0, zeta0
    p.eta = eta0 + L * M / 2
    p.zeta = zeta0
    p.sigmaVecHat = sigma * np.ones(L)
    p.etaVec = (eta0 + M / 2) * np.ones(L)
    p.zetaVec = zeta0 * np.ones(L)
    p.YHat = p.BHat @ p.AHat.T if materialize_yhat else None
    p.trYTY = traceXTY(Y, Y)
    return p


def spread_v(v, M, reference_compat=True):
    """The H-vector -> M*H expansion of src/vbmf_sparse.jl:221.

    reference_compat=True reproduces `repeat(v, inner=M-1)` after the first H entries (QS1):
    position p >= H uses v[(p-H) // (M-1)].  False = the consistent tiling v[p % H].
    """
    H = v.shape[0]
    if reference_compat:
        return np.concatenate([v, np.repeat(v, M - 1)])
    return np.tile(v, M)


def sparse_updateA(Y, p, full_cov=False, reference_compat=True, diag_var=False):
    """src/vbmf_sparse.jl:176-247, all four (full_cov, diag_var) branches."""
    L, M, H = p.L, p.M, p.H
    if full_cov and diag_var:                                       # :180-182, :192-193 (sigma enters to the FIRST power here)
        K = p.BHat.T @ (p.sigmaVecHat[:, None] * p.BHat) + L * np.mean(p.sigmaVecHat) * p.SigmaB
        p.invSigmaATVec = np.kron(np.eye(M), K) + np.diag(p.CA)
        p.SigmaATVec = np.linalg.inv(p.invSigmaATVec)
        p.diagSigmaATVec = np.diag(p.SigmaATVec).copy()
        BtY = (p.BHat * p.sigmaVecHat[:, None]).T @ Y               # B' diag(sigmaVec) Y
        p.ATVecHat = p.SigmaATVec @ BtY.T.reshape(M * H)            # :193 (no sigmaHat factor)
        p.SigmaA = np.zeros((H, H))
        for m in range(M):
            p.SigmaA += p.SigmaATVec[m * H:(m + 1) * H, m * H:(m + 1) * H]
    elif not full_cov and diag_var:                                   # :207-212, 229-230 (heteroscedastic rows)
        sB = p.BHat * p.sigmaVecHat[:, None]
        v = np.sum(sB * sB, axis=0) + L * np.mean(p.sigmaVecHat) * np.diag(p.SigmaB)   # :211 (sigma enters squared)
        prec = spread_v(v, M, reference_compat) + p.CA
        p.diagSigmaATVec = 1.0 / prec
        BtY = sB.T @ Y                                              # B' diag(sigmaVec) Y
        p.ATVecHat = p.diagSigmaATVec * BtY.T.reshape(M * H)        # :230 (no sigmaHat factor)
        p.SigmaA = np.diag(p.diagSigmaATVec.reshape(M, H).sum(axis=0))
    elif full_cov:                                                  # :178-202
        K = p.sigmaHat * (p.BHat.T @ p.BHat + L * p.SigmaB)
        p.invSigmaATVec = np.kron(np.eye(M), K) + np.diag(p.CA)
        p.SigmaATVec = np.linalg.inv(p.invSigmaATVec)
        p.diagSigmaATVec = np.diag(p.SigmaATVec).copy()
        BtY = p.BHat.T @ Y                                          # H x M; vec = column-major
        p.ATVecHat = p.sigmaHat * (p.SigmaATVec @ BtY.T.reshape(M * H))
        p.SigmaA = np.zeros((H, H))
        for m in range(M):
            p.SigmaA += p.SigmaATVec[m * H:(m + 1) * H, m * H:(m + 1) * H]
    else:                                                           # :204-240
        v = p.sigmaHat * np.sum(p.BHat * p.BHat, axis=0) + L * np.diag(p.SigmaB)   # :217 (QS2)
        prec = spread_v(v, M, reference_compat) + p.CA             # :221-223
        p.diagSigmaATVec = 1.0 / prec                               # :226
        BtY = p.BHat.T @ Y
        p.ATVecHat = p.sigmaHat * p.diagSigmaATVec * BtY.T.reshape(M * H)           # :232
        p.SigmaA = np.diag(p.diagSigmaATVec.reshape(M, H).sum(axis=0))              # :236-239
    p.AHat = p.ATVecHat.reshape(M, H).copy()                        # :244
    _mask(p.AHat, p.labels, p.H1)                                   # :245
    p.ATVecHat = p.AHat.reshape(M * H).copy()                       # :246


def sparse_updateB(Y, p, diag_var=False):
    """src/vbmf_sparse.jl:254-268."""
    if diag_var:                                                    # :256-261
        p.SigmaB = np.linalg.inv(np.diag(p.CB) + np.mean(p.sigmaVecHat) * (p.AHat.T @ p.AHat + p.SigmaA))
        p.BHat = p.sigmaVecHat[:, None] * ((Y @ p.AHat) @ p.SigmaB)
        return
    p.SigmaB = np.linalg.inv(np.diag(p.CB) + p.sigmaHat * (p.AHat.T @ p.AHat + p.SigmaA))
    p.BHat = p.sigmaHat * ((Y @ p.AHat) @ p.SigmaB)


def sparse_updateCA(p):
    """src/vbmf_sparse.jl:284-288."""
    p.beta = p.beta0 + 0.5 * (p.ATVecHat * p.ATVecHat + p.diagSigmaATVec)
    p.CA = p.alpha / p.beta


def sparse_updateCB(p):
    """src/vbmf_sparse.jl:295-300."""
    p.delta = p.delta0 + 0.5 * np.sum(p.BHat * p.BHat, axis=0) + 0.5 * np.diag(p.SigmaB)
    p.CB = p.gamma / p.delta


def sparse_updateSigma(Y, p, diag_var=False):
    """src/vbmf_sparse.jl:308-321."""
    if diag_var:                                                    # :309-315, row by row
        G = p.AHat.T @ p.AHat + p.SigmaA
        Q = Y @ p.AHat
        quad = np.einsum("lh,hk,lk->l", p.BHat, G, p.BHat) + np.sum(G * p.SigmaB)    # tr(G (b b' + SigmaB))
        p.zetaVec = p.zeta0 + 0.5 * np.sum(Y * Y, axis=1) - np.sum(Q * p.BHat, axis=1) + 0.5 * quad
        p.sigmaVecHat = p.etaVec / p.zetaVec
        return
    p.zeta = (p.zeta0 + 0.5 * p.trYTY - traceXTY(p.BHat, Y @ p.AHat)
              + 0.5 * traceXTY(p.AHat.T @ p.AHat + p.SigmaA, p.BHat.T @ p.BHat + p.L * p.SigmaB))
    p.sigmaHat = p.eta / p.zeta


def vbmf_sparse_(Y, p, niter, eps=1e-6, full_cov=False, est_cb=True, reference_compat=True, trace=None, diag_var=False):
    """vbmf_sparse! -- src/vbmf_sparse.jl:344-410.  Returns (d, iterations)."""
    old = p.BHat.copy()
    d = eps + 1.0
    i = 1
    while i <= niter and d > eps:
        sparse_updateA(Y, p, full_cov=full_cov, reference_compat=reference_compat, diag_var=diag_var)
        sparse_updateB(Y, p, diag_var=diag_var)
        sparse_updateCA(p)
        if est_cb:
            sparse_updateCB(p)
        sparse_updateSigma(Y, p, diag_var=diag_var)
        d = delta(p.BHat, old)
        old = p.BHat.copy()
        if trace is not None:
            trace.append((d, p.sigmaHat))
        i += 1
    return d, i - 1


def lowerBound(Y, p, clamp=True):
    """src/vbmf_sparse.jl:435-471, verbatim quirks QS4; H(B) restated as L*logdet(SigmaB) (QS5)."""
    from scipy.special import gammaln
    L_, M, H = p.L, p.M, p.H
    MH = p.ATVecHat.shape[0]
    Lb = 0.0
    Lb += -L_ * M / 2 * LN2PI + L_ * M / 2 * gammaELn(p.eta, p.zeta)
    Lb += -p.sigmaHat / 2 * (p.trYTY - 2 * traceXTY(p.BHat, Y @ p.AHat)
                             + traceXTY(p.AHat.T @ p.AHat + p.SigmaA, p.BHat.T @ p.BHat + L_ * p.SigmaB))
    eln_ca = gammaELn(p.alpha, p.beta)
    Lb += -MH / 2 * LN2PI + 0.5 * float(np.sum(eln_ca))
    Lb += -0.5 * float(p.CA @ (p.ATVecHat ** 2 + p.diagSigmaATVec))
    Lb += -L_ * H / 2 * LN2PI
    eln_cb = gammaELn(p.gamma, p.delta)
    Lb += L_ / 2 * float(np.sum(eln_cb))
    Lb += -0.5 * traceXTY(np.diag(p.CB), p.BHat.T @ p.BHat + L_ * p.SigmaB)
    Lb += p.eta0 * math.log(p.zeta0) - gammaln(p.eta0)
    Lb += (p.eta0 - 1) * gammaELn(p.eta, p.zeta) - p.zeta0 * p.sigmaHat
    Lb += MH * (p.alpha0 * math.log(p.beta0) - gammaln(p.alpha0))
    Lb += (p.alpha0 - 1) * float(np.sum(eln_ca))
    Lb += -p.beta0 * float(np.sum(p.CA))
    Lb += H * (p.gamma0 * math.log(p.delta0) - gammaln(p.gamma0))
    Lb += (p.gamma0 - 1) * float(np.sum(eln_cb))
    Lb += -p.gamma0 * float(np.sum(p.CB))                           # sic: gamma0 (QS4)
    Lb += normalEntropy_diag(p.diagSigmaATVec)
    sgn, ld = np.linalg.slogdet(p.SigmaB)
    logdet_kron = L_ * ld if sgn > 0 else -math.inf                 # det(kron(SigmaB, I_L)) = det(SigmaB)^L
    Lb += normalEntropy_matrix_logdet(L_ * H, logdet_kron, clamp=clamp)
    Lb += float(gammaEntropy(p.eta, p.zeta))
    Lb += float(np.sum(gammaEntropy(p.alpha, p.beta)))
    Lb += float(np.sum(gammaEntropy(p.gamma, p.delta)))
    return float(Lb)


def lowerBoundTrimmed(Y, p_in, trim=1e-1, clamp=True):
    """src/vbmf_sparse.jl:478-489 (the grouped models' copies: src/vbmf_dual.jl:606-617, src/vbmf_trial.jl:687-698): the
    entries of vec(A') with abs(ATVecHat) <= trim leave ATVecHat, beta, CA, diagSigmaATVec and MH; AHat (hence Y*AHat,
    AHat'AHat) and every per-group field (beta0, CA0, ...) stay whole -- exactly the fields the reference reassigns --
    then lowerBound of the type."""
    import copy as _copy
    p = _copy.copy(p_in)
    keep = np.abs(p.ATVecHat) > trim                                # :481
    p.ATVecHat = p.ATVecHat[keep]
    p.MH = int(p.ATVecHat.shape[0])
    p.beta = p.beta[keep]
    p.CA = p.CA[keep]
    p.diagSigmaATVec = p.diagSigmaATVec[keep]
    if isinstance(p, vbmf_trial_parameters):
        return lowerBound_trial(Y, p, clamp=clamp)
    if isinstance(p, vbmf_dual_parameters):
        return lowerBound_dual(Y, p, clamp=clamp)
    return lowerBound(Y, p, clamp=clamp)


# ----------------------------------------------------------------------------------------------
# Two-group ARD variant  (src/vbmf_dual.jl), diagonal branch (full_cov=false)
#
# A = [A0 A1] with H0 + H1 = H columns; each group has its own Gamma hyper-prior on the element-wise precisions
# (alpha00, beta00 / alpha01, beta01) which `est_priors=true` re-fits every sweep by maximising the bound
# (src/vbmf_dual.jl:393-434).  updateA!/updateB!/updateCB!/updateSigma! are the sparse model's bodies without the
# label mask (:216-306, 358-386).  The fit of alpha00/alpha01 calls `fzero` of Roots.jl -- a third-party package that
# is neither vendored nor pinned (it is absent from the reference's REQUIRE): PARITY UNPINNED for est_priors=true.
# Restated as the exact root of the same monotone function on the same bracket [1e-10, 1e10]; like the reference's
# `try ... end`, a bracket without a sign change leaves the value unchanged.
# ----------------------------------------------------------------------------------------------
@dataclass
class vbmf_dual_parameters(vbmf_sparse_parameters):
    """src/vbmf_dual.jl:59-112.  Field names of the reference; `alpha`/`beta`/`CA` are the interleaved
    (m, h) vectors of :146-165, CA0/CA1/beta0/beta1 the per-group ones (index (m-1)*H_g + h)."""
    H0: int = 0
    A0Hat: Optional[np.ndarray] = None
    A1Hat: Optional[np.ndarray] = None
    CA0: Optional[np.ndarray] = None
    CA1: Optional[np.ndarray] = None
    alpha00: float = 1e-10
    beta00: float = 1e-10
    alpha01: float = 1e-10
    beta01: float = 1e-10
    alpha1: float = 0.0
    beta1: Optional[np.ndarray] = None
    # NB: in this type `alpha0` is the POSTERIOR shape of group 0 and `beta0` its M*H0 rate vector (:31-32 of the
    # reference's docstring); the scalar priors are alpha00/beta00.


def _dual_split(vecMH, M, H, H0):
    a = vecMH.reshape(M, H)
    return a[:, :H0].reshape(M * H0).copy(), a[:, H0:].reshape(M * (H - H0)).copy()


def _dual_join(v0, v1, M, H, H0):
    return np.concatenate([v0.reshape(M, H0), v1.reshape(M, H - H0)], axis=1).reshape(M * H)


def vbmf_dual_init(Y, H, H0, ca=1.0, alpha0=1e-10, beta0=1e-10, cb=1.0, gamma0=1e-10, delta0=1e-10,
                   sigma=1.0, eta0=1e-10, zeta0=1e-10, rng=None, materialize_yhat=True):
    """src/vbmf_dual.jl:122-193 (the two eye(MH) allocations of :141,143 are skipped: diagonal branch only)."""
    if H < H0:
        raise ValueError("H must be at least H0!")                       # :126-128
    rng = np.random.default_rng(0) if rng is None else rng
    p = vbmf_dual_parameters()
    L, M = Y.shape
    H1 = H - H0
    p.L, p.M, p.H, p.MH, p.H0, p.H1 = L, M, H, M * H, H0, H1
    p.AHat = rng.standard_normal((M, H))
    p.ATVecHat = p.AHat.reshape(M * H).copy()
    p.diagSigmaATVec = np.ones(M * H)
    p.SigmaA = np.zeros((H, H))
    p.A0Hat, p.A1Hat = p.AHat[:, :H0].copy(), p.AHat[:, H0:].copy()
    p.BHat = rng.standard_normal((L, H))
    p.SigmaB = np.zeros((H, H))
    p.CA0, p.CA1 = ca * np.ones(M * H0), ca * np.ones(M * H1)
    p.CA = _dual_join(p.CA0, p.CA1, M, H, H0)
    p.alpha00 = p.alpha01 = alpha0
    p.beta00 = p.beta01 = beta0
    p.alpha0 = alpha0 + 0.5
    p.alpha1 = alpha0 + 0.5
    p.beta0, p.beta1 = beta0 * np.ones(M * H0), beta0 * np.ones(M * H1)
    p.alpha = np.array([p.alpha0, p.alpha1])
    p.beta = _dual_join(p.beta0, p.beta1, M, H, H0)
    p.CB = cb * np.ones(H)
    p.gamma0, p.delta0 = gamma0, delta0
    p.gamma = gamma0 + L / 2
    p.delta = delta0 * np.ones(H)
    p.sigmaHat = float(sigma)
    p.eta0, p.zeta0 = eta0, zeta0
    p.eta = eta0 + L * M / 2
    p.zeta = zeta0
    p.sigmaVecHat = sigma * np.ones(L)
    p.etaVec = (eta0 + M / 2) * np.ones(L)
    p.zetaVec = zeta0 * np.ones(L)
    p.YHat = p.BHat @ p.AHat.T if materialize_yhat else None
    p.trYTY = traceXTY(Y, Y)
    return p


def dual_updateA(Y, p, reference_compat=True, diag_var=False, full_cov=False):
    """src/vbmf_dual.jl:216-284: the sparse model's updateA! (either branch), no label mask, then the A0/A1 views."""
    sparse_updateA(Y, p, full_cov=full_cov, reference_compat=reference_compat, diag_var=diag_var)
    p.A0Hat, p.A1Hat = p.AHat[:, :p.H0].copy(), p.AHat[:, p.H0:].copy()


def dual_updateCA(p):
    """src/vbmf_dual.jl:322-351."""
    M, H, H0 = p.M, p.H, p.H0
    p.alpha0 = p.alpha00 + 0.5
    p.alpha1 = p.alpha01 + 0.5
    q0, q1 = _dual_split(p.ATVecHat * p.ATVecHat + p.diagSigmaATVec, M, H, H0)
    p.beta0 = p.beta00 + 0.5 * q0
    p.beta1 = p.beta01 + 0.5 * q1
    p.CA0 = p.alpha0 / p.beta0
    p.CA1 = p.alpha1 / p.beta1
    p.CA = _dual_join(p.CA0, p.CA1, M, H, H0)
    p.alpha = np.array([p.alpha0, p.alpha1])
    p.beta = _dual_join(p.beta0, p.beta1, M, H, H0)


def _dual_fit_shape(n, log_rate_prior, shape_post, rates, current):
    """Root of  n*log(beta0g) - n*digamma(x) + sum_i gammaELn(shape_post, rates_i)  on [1e-10, 1e10]
    (src/vbmf_dual.jl:394-400, 418-424)."""
    from scipy.optimize import brentq
    from scipy.special import digamma
    if n == 0:
        return current
    s = float(np.sum(gammaELn(shape_post, rates)))
    f = lambda x: n * log_rate_prior - n * float(digamma(x)) + s
    lo, hi = 1e-10, 1e10
    flo, fhi = f(lo), f(hi)
    if not (np.isfinite(flo) and np.isfinite(fhi)) or flo * fhi > 0:
        return current                                                   # fzero throws, `try ... end` swallows it
    return float(brentq(f, lo, hi, xtol=1e-300, rtol=4 * np.finfo(float).eps, maxiter=500))


def dual_updatePriors(p):
    """updateAlpha00!, updateAlpha01!, updateBeta00!, updateBeta01! in the order of src/vbmf_dual.jl:491-495."""
    n0, n1 = p.M * p.H0, p.M * p.H1
    p.alpha00 = _dual_fit_shape(n0, math.log(p.beta00), p.alpha0, p.beta0, p.alpha00)
    p.alpha01 = _dual_fit_shape(n1, math.log(p.beta01), p.alpha1, p.beta1, p.alpha01)
    if n0:
        p.beta00 = n0 * p.alpha00 / float(np.sum(p.CA0))                 # :408-410
    if n1:
        p.beta01 = n1 * p.alpha01 / float(np.sum(p.CA1))                 # :432-434


def vbmf_dual_(Y, p, niter, eps=1e-6, est_cb=True, est_priors=True, reference_compat=True, diag_var=False, trace=None,
               full_cov=False):
    """vbmf_dual! -- src/vbmf_dual.jl:455-530 (full_cov=false, convergence on BHat).  Returns (d, iterations)."""
    old = p.BHat.copy()
    d = eps + 1.0
    i = 1
    while i <= niter and d > eps:
        dual_updateA(Y, p, reference_compat=reference_compat, diag_var=diag_var, full_cov=full_cov)
        sparse_updateB(Y, p, diag_var=diag_var)                          # :292-306
        dual_updateCA(p)
        if est_cb:
            sparse_updateCB(p)                                           # :358-363
        sparse_updateSigma(Y, p, diag_var=diag_var)                      # :370-386
        if est_priors:
            dual_updatePriors(p)
        d = delta(p.BHat, old)
        old = p.BHat.copy()
        if trace is not None:
            trace.append((d, p.sigmaHat, p.alpha00, p.beta00, p.alpha01, p.beta01))
        i += 1
    return d, i - 1


def vbls_dual_(Y, p, niter, reference_compat=True):
    """examples/mil_util.jl:190-193, vbmf_dual_parameters branch (full_cov=false, diag_var=false)."""
    for _ in range(niter):
        dual_updateA(Y, p, reference_compat=reference_compat)
        dual_updateCA(p)
        sparse_updateSigma(Y, p)
    return p.AHat


def lowerBound_dual(Y, p, clamp=True):
    """src/vbmf_dual.jl:556-599 (homoscedastic), same sic's as the sparse bound; H(B) as L*logdet(SigmaB)."""
    from scipy.special import gammaln
    L_, M, H, H0, H1 = p.L, p.M, p.H, p.H0, p.H1
    MH = p.ATVecHat.shape[0]
    e0 = float(np.sum(gammaELn(p.alpha0, p.beta0))) if M * H0 else 0.0
    e1 = float(np.sum(gammaELn(p.alpha1, p.beta1))) if M * H1 else 0.0
    Lb = 0.0
    Lb += -L_ * M / 2 * LN2PI + L_ * M / 2 * gammaELn(p.eta, p.zeta)
    Lb += -p.sigmaHat / 2 * (p.trYTY - 2 * traceXTY(p.BHat, Y @ p.AHat)
                             + traceXTY(p.AHat.T @ p.AHat + p.SigmaA, p.BHat.T @ p.BHat + L_ * p.SigmaB))
    Lb += -MH / 2 * LN2PI + 0.5 * e0 + 0.5 * e1                                        # :564-565
    Lb += -0.5 * float(p.CA @ (p.ATVecHat ** 2 + p.diagSigmaATVec))                    # :566
    Lb += -L_ * H / 2 * LN2PI
    eln_cb = gammaELn(p.gamma, p.delta)
    Lb += L_ / 2 * float(np.sum(eln_cb))
    Lb += -0.5 * traceXTY(np.diag(p.CB), p.BHat.T @ p.BHat + L_ * p.SigmaB)
    Lb += p.eta0 * math.log(p.zeta0) - gammaln(p.eta0)
    Lb += (p.eta0 - 1) * gammaELn(p.eta, p.zeta) - p.zeta0 * p.sigmaHat
    Lb += M * H0 * (p.alpha00 * math.log(p.beta00) - gammaln(p.alpha00))               # :575-577
    Lb += (p.alpha00 - 1) * e0 - p.beta00 * float(np.sum(p.CA0))
    Lb += M * H1 * (p.alpha01 * math.log(p.beta01) - gammaln(p.alpha01))               # :579-581
    Lb += (p.alpha01 - 1) * e1 - p.beta01 * float(np.sum(p.CA1))
    Lb += H * (p.gamma0 * math.log(p.delta0) - gammaln(p.gamma0))
    Lb += (p.gamma0 - 1) * float(np.sum(eln_cb))
    Lb += -p.gamma0 * float(np.sum(p.CB))                                              # sic: gamma0 (:585)
    Lb += normalEntropy_diag(p.diagSigmaATVec)
    sgn, ld = np.linalg.slogdet(p.SigmaB)
    logdet_kron = L_ * ld if sgn > 0 else -math.inf
    Lb += normalEntropy_matrix_logdet(L_ * H, logdet_kron, clamp=clamp)
    Lb += float(gammaEntropy(p.eta, p.zeta))
    Lb += float(np.sum(gammaEntropy(p.alpha0, p.beta0))) if M * H0 else 0.0            # :593-596
    Lb += float(np.sum(gammaEntropy(p.alpha1, p.beta1))) if M * H1 else 0.0
    Lb += float(np.sum(gammaEntropy(p.gamma, p.delta)))
    return float(Lb)


# ----------------------------------------------------------------------------------------------
# Three-group ARD variant  (src/vbmf_trial.jl), diagonal branch (full_cov=false)
#
# A = [A1 [A2; A3]]: A1 = the first H0 columns (all rows), A2 / A3 = the other H1 columns of rows 1..M0 / M0+1..M, each
# block with its own hyper-prior (alpha0g, beta0g).  updateA!/updateB!/updateCB!/updateSigma! are again the sparse
# model's bodies (:250-341, 407-435); updateCA! and the fits are the two-group model's with one more group
# (:357-400, 442-507).  Same `fzero` caveat: PARITY UNPINNED for est_priors=true.
# ----------------------------------------------------------------------------------------------
@dataclass
class vbmf_trial_parameters(vbmf_sparse_parameters):
    """src/vbmf_trial.jl:68-131.  `alpha`/`beta`/`CA`: the interleaved (m, h) vectors of :178-211."""
    M0: int = 0
    M1: int = 0
    H0: int = 0
    A1Hat: Optional[np.ndarray] = None
    A2Hat: Optional[np.ndarray] = None
    A3Hat: Optional[np.ndarray] = None
    CA1: Optional[np.ndarray] = None
    CA2: Optional[np.ndarray] = None
    CA3: Optional[np.ndarray] = None
    alpha01: float = 1e-10
    beta01: float = 1e-10
    alpha02: float = 1e-10
    beta02: float = 1e-10
    alpha03: float = 1e-10
    beta03: float = 1e-10
    alpha1: float = 0.0
    alpha2: float = 0.0
    alpha3: float = 0.0
    beta1: Optional[np.ndarray] = None
    beta2: Optional[np.ndarray] = None
    beta3: Optional[np.ndarray] = None


def _trial_split(vecMH, M, H, H0, M0):
    a = vecMH.reshape(M, H)
    H1 = H - H0
    return (a[:, :H0].reshape(M * H0).copy(), a[:M0, H0:].reshape(M0 * H1).copy(), a[M0:, H0:].reshape((M - M0) * H1).copy())


def _trial_join(v1, v2, v3, M, H, H0, M0):
    H1 = H - H0
    right = np.concatenate([v2.reshape(M0, H1), v3.reshape(M - M0, H1)], axis=0)
    return np.concatenate([v1.reshape(M, H0), right], axis=1).reshape(M * H)


def vbmf_trial_init(Y, H, H0, M0, ca=1.0, alpha0=1e-10, beta0=1e-10, cb=1.0, gamma0=1e-10, delta0=1e-10,
                    sigma=1.0, eta0=1e-10, zeta0=1e-10, rng=None, materialize_yhat=True):
    """src/vbmf_trial.jl:139-226 (without the two eye(MH) allocations of :162,164)."""
    if H < H0:
        raise ValueError("H must be at least H0!")                       # :143-145
    rng = np.random.default_rng(0) if rng is None else rng
    p = vbmf_trial_parameters()
    L, M = Y.shape
    H1, M1 = H - H0, M - M0
    p.L, p.M, p.H, p.MH, p.H0, p.H1, p.M0, p.M1 = L, M, H, M * H, H0, H1, M0, M1
    p.AHat = rng.standard_normal((M, H))
    p.ATVecHat = p.AHat.reshape(M * H).copy()
    p.diagSigmaATVec = np.ones(M * H)
    p.SigmaA = np.zeros((H, H))
    p.A1Hat, p.A2Hat, p.A3Hat = p.AHat[:, :H0].copy(), p.AHat[:M0, H0:].copy(), p.AHat[M0:, H0:].copy()
    p.BHat = rng.standard_normal((L, H))
    p.SigmaB = np.zeros((H, H))
    p.CA1, p.CA2, p.CA3 = ca * np.ones(M * H0), ca * np.ones(M0 * H1), ca * np.ones(M1 * H1)
    p.CA = _trial_join(p.CA1, p.CA2, p.CA3, M, H, H0, M0)
    p.alpha01 = p.alpha02 = p.alpha03 = alpha0
    p.beta01 = p.beta02 = p.beta03 = beta0
    p.alpha1 = p.alpha2 = p.alpha3 = alpha0 + 0.5
    p.beta1, p.beta2, p.beta3 = beta0 * np.ones(M * H0), beta0 * np.ones(M0 * H1), beta0 * np.ones(M1 * H1)
    p.alpha = np.array([p.alpha1, p.alpha2, p.alpha3])
    p.beta = _trial_join(p.beta1, p.beta2, p.beta3, M, H, H0, M0)
    p.CB = cb * np.ones(H)
    p.gamma0, p.delta0 = gamma0, delta0
    p.gamma = gamma0 + L / 2
    p.delta = delta0 * np.ones(H)
    p.sigmaHat = float(sigma)
    p.eta0, p.zeta0 = eta0, zeta0
    p.eta = eta0 + L * M / 2
    p.zeta = zeta0
    p.sigmaVecHat = sigma * np.ones(L)
    p.etaVec = (eta0 + M / 2) * np.ones(L)
    p.zetaVec = zeta0 * np.ones(L)
    p.YHat = p.BHat @ p.AHat.T if materialize_yhat else None
    p.trYTY = traceXTY(Y, Y)
    return p


def trial_updateA(Y, p, reference_compat=True, diag_var=False, full_cov=False):
    """src/vbmf_trial.jl:250-320: the sparse model's updateA! (either branch), no label mask, then the A1/A2/A3 views."""
    sparse_updateA(Y, p, full_cov=full_cov, reference_compat=reference_compat, diag_var=diag_var)
    p.A1Hat, p.A2Hat, p.A3Hat = p.AHat[:, :p.H0].copy(), p.AHat[:p.M0, p.H0:].copy(), p.AHat[p.M0:, p.H0:].copy()


def trial_updateCA(p):
    """src/vbmf_trial.jl:357-400."""
    M, H, H0, M0 = p.M, p.H, p.H0, p.M0
    p.alpha1, p.alpha2, p.alpha3 = p.alpha01 + 0.5, p.alpha02 + 0.5, p.alpha03 + 0.5
    q1, q2, q3 = _trial_split(p.ATVecHat * p.ATVecHat + p.diagSigmaATVec, M, H, H0, M0)
    p.beta1 = p.beta01 + 0.5 * q1
    p.beta2 = p.beta02 + 0.5 * q2
    p.beta3 = p.beta03 + 0.5 * q3
    p.CA1, p.CA2, p.CA3 = p.alpha1 / p.beta1, p.alpha2 / p.beta2, p.alpha3 / p.beta3
    p.CA = _trial_join(p.CA1, p.CA2, p.CA3, M, H, H0, M0)
    p.alpha = np.array([p.alpha1, p.alpha2, p.alpha3])
    p.beta = _trial_join(p.beta1, p.beta2, p.beta3, M, H, H0, M0)


def trial_updatePriors(p):
    """updateAlpha01!..03!, then updateBeta01!..03! (src/vbmf_trial.jl:565-572, 442-507)."""
    n = (p.M * p.H0, p.M0 * p.H1, p.M1 * p.H1)
    p.alpha01 = _dual_fit_shape(n[0], math.log(p.beta01), p.alpha1, p.beta1, p.alpha01)
    p.alpha02 = _dual_fit_shape(n[1], math.log(p.beta02), p.alpha2, p.beta2, p.alpha02)
    p.alpha03 = _dual_fit_shape(n[2], math.log(p.beta03), p.alpha3, p.beta3, p.alpha03)
    if n[0]:
        p.beta01 = n[0] * p.alpha01 / float(np.sum(p.CA1))
    if n[1]:
        p.beta02 = n[1] * p.alpha02 / float(np.sum(p.CA2))
    if n[2]:
        p.beta03 = n[2] * p.alpha03 / float(np.sum(p.CA3))


def vbmf_trial_(Y, p, niter, eps=1e-6, est_cb=True, est_priors=True, reference_compat=True, trace=None, diag_var=False,
                full_cov=False):
    """vbmf_trial! -- src/vbmf_trial.jl:528-604 (full_cov=false).  Returns (d, iterations)."""
    old = p.BHat.copy()
    d = eps + 1.0
    i = 1
    while i <= niter and d > eps:
        trial_updateA(Y, p, reference_compat=reference_compat, diag_var=diag_var, full_cov=full_cov)
        sparse_updateB(Y, p, diag_var=diag_var)                          # :327-341
        trial_updateCA(p)
        if est_cb:
            sparse_updateCB(p)                                           # :407-412
        sparse_updateSigma(Y, p, diag_var=diag_var)                      # :419-435
        if est_priors:
            trial_updatePriors(p)
        d = delta(p.BHat, old)
        old = p.BHat.copy()
        if trace is not None:
            trace.append((d, p.sigmaHat, p.alpha01, p.beta01, p.alpha02, p.beta02, p.alpha03, p.beta03))
        i += 1
    return d, i - 1


def vbls_trial_(Y, p, niter, reference_compat=True):
    """examples/mil_util.jl:194-197, vbmf_trial_parameters branch."""
    for _ in range(niter):
        trial_updateA(Y, p, reference_compat=reference_compat)
        trial_updateCA(p)
        sparse_updateSigma(Y, p)
    return p.AHat


def lowerBound_trial(Y, p, clamp=True):
    """src/vbmf_trial.jl:630-680; H(B) as L*logdet(SigmaB)."""
    from scipy.special import gammaln
    L_, M, H = p.L, p.M, p.H
    MH = p.ATVecHat.shape[0]
    groups = [(p.M * p.H0, p.alpha01, p.beta01, p.alpha1, p.beta1, p.CA1),
              (p.M0 * p.H1, p.alpha02, p.beta02, p.alpha2, p.beta2, p.CA2),
              (p.M1 * p.H1, p.alpha03, p.beta03, p.alpha3, p.beta3, p.CA3)]
    eln = [float(np.sum(gammaELn(ap, b))) if n else 0.0 for n, _, _, ap, b, _ in groups]
    Lb = 0.0
    Lb += -L_ * M / 2 * LN2PI + L_ * M / 2 * gammaELn(p.eta, p.zeta)
    Lb += -p.sigmaHat / 2 * (p.trYTY - 2 * traceXTY(p.BHat, Y @ p.AHat)
                             + traceXTY(p.AHat.T @ p.AHat + p.SigmaA, p.BHat.T @ p.BHat + L_ * p.SigmaB))
    Lb += -MH / 2 * LN2PI + 0.5 * sum(eln)                                             # :637-639
    Lb += -0.5 * float(p.CA @ (p.ATVecHat ** 2 + p.diagSigmaATVec))
    Lb += -L_ * H / 2 * LN2PI
    eln_cb = gammaELn(p.gamma, p.delta)
    Lb += L_ / 2 * float(np.sum(eln_cb))
    Lb += -0.5 * traceXTY(np.diag(p.CB), p.BHat.T @ p.BHat + L_ * p.SigmaB)
    Lb += p.eta0 * math.log(p.zeta0) - gammaln(p.eta0)
    Lb += (p.eta0 - 1) * gammaELn(p.eta, p.zeta) - p.zeta0 * p.sigmaHat
    for (n, a0, b0, _, _, ca), e in zip(groups, eln):                                  # :649-659
        if n:
            Lb += n * (a0 * math.log(b0) - gammaln(a0)) + (a0 - 1) * e - b0 * float(np.sum(ca))
    Lb += H * (p.gamma0 * math.log(p.delta0) - gammaln(p.gamma0))
    Lb += (p.gamma0 - 1) * float(np.sum(eln_cb))
    Lb += -p.gamma0 * float(np.sum(p.CB))                                              # sic: gamma0 (:663)
    Lb += normalEntropy_diag(p.diagSigmaATVec)
    sgn, ld = np.linalg.slogdet(p.SigmaB)
    logdet_kron = L_ * ld if sgn > 0 else -math.inf
    Lb += normalEntropy_matrix_logdet(L_ * H, logdet_kron, clamp=clamp)
    Lb += float(gammaEntropy(p.eta, p.zeta))
    for n, _, _, ap, b, _ in groups:                                                   # :671-676
        if n:
            Lb += float(np.sum(gammaEntropy(ap, b)))
    Lb += float(np.sum(gammaEntropy(p.gamma, p.delta)))
    return float(Lb)


# ----------------------------------------------------------------------------------------------
# synthetic data (generalises toy_matrix, examples/toy_data.jl:7-18)
# ----------------------------------------------------------------------------------------------
def toy_matrix(L, M, H, std, rng):
    B = rng.standard_normal((L, H))
    A = np.zeros((M, H))
    A[np.arange(M), rng.integers(0, H, size=M)] = 1.0
    Y = B @ A.T + std * rng.standard_normal((L, M))
    return Y, A, B
