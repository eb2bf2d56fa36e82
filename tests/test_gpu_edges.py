"""Edge cases of the drop-in boundary, through the C ABI on the GPU: degenerate and ragged shapes (one row, one
column, rank 1, sizes straddling the 32-wide tiles and the k-step), zero sweeps, immediate termination, re-entrancy,
independence of handles, leading dimensions, NaN data, and the error behaviour of every argument check."""
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


def _oracle_run(Y, H, seed, n, **kw):
    po = O.vbmf_init(Y, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(seed), materialize_yhat=False, **kw)
    init = (po.AHat.copy(), po.BHat.copy())
    O.vbmf_(Y, po, n, eps=0.0, est_covs=True, est_var=True)
    return po, init


@pytest.mark.parametrize("L,M,H", [(1, 1, 1), (1, 40, 1), (40, 1, 1), (2, 3, 2), (31, 33, 1), (33, 31, 3), (32, 32, 32),
                                   (65, 17, 5), (17, 65, 16), (64, 64, 33), (100, 7, 7), (7, 100, 7)])
def test_ragged_and_degenerate_shapes(pkg, L, M, H):
    """Two sweeps on every shape, fp32 path, against the oracle (tolerance as test_each_update_f32 x sweeps)."""
    rng = np.random.default_rng(1000 + L * 7 + M)
    Y = rng.standard_normal((L, M)).astype(np.float32).astype(np.float64)
    po, (A0, B0) = _oracle_run(Y, H, 5, 2)
    z = np.zeros((H, H))
    with pkg.capi.Context(L, M, H, y_dtype=pkg.VBMF_Y_F32) as c:
        c.set_Y(Y)
        assert np.array_equal(c.get_Y(), Y)
        c.set_state(A0, B0, z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
        it, d, _ = c.run(2, eps=0.0, est_covs=True, est_var=True)
        s = c.get_state()
        yh = c.YHat()
    errs = dict(A=relF(s["AHat"], po.AHat), B=relF(s["BHat"], po.BHat), SA=relF(s["SigmaA"], po.SigmaA),
                SB=relF(s["SigmaB"], po.SigmaB), s2=abs(s["sigma2"] - po.sigma2) / po.sigma2)
    report(f"edge shape {L}x{M} H={H}: " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert it == 2 and max(errs.values()) < 5e-4, errs
    assert relF(yh, po.BHat @ po.AHat.T) < 5e-4


def test_zero_sweeps_and_immediate_stop(pkg):
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((50, 30))
    A0, B0 = rng.standard_normal((30, 4)), rng.standard_normal((50, 4))
    z = np.zeros((4, 4))
    with pkg.capi.Context(50, 30, 4, y_dtype=pkg.VBMF_Y_F32) as c:
        c.set_Y(Y)
        c.set_state(A0, B0, z, z, 0.1 * np.ones(4), 0.1 * np.ones(4), 0.1)
        it, d, _ = c.run(0, eps=1e-6, est_covs=True, est_var=True)         # niter = 0: nothing runs, d = eps + 1 (:189)
        s = c.get_state()
        assert it == 0 and d == 1e-6 + 1.0
        assert relF(s["AHat"], A0) < 1e-7 and relF(s["BHat"], B0) < 1e-7 and s["sigma2"] == pytest.approx(0.1)
        it, d, _ = c.run(50, eps=1e9, est_covs=True, est_var=True)         # d <= eps after the first sweep: exactly one
        assert it == 1 and np.isfinite(d)
        s1 = c.get_state()
        it, d2, _ = c.run(1, eps=0.0, est_covs=True, est_var=True)         # re-entrancy: continue from where it stopped
        assert it == 1
    po = O.vbmf_parameters(); po.L, po.M, po.H = 50, 30, 4
    po.AHat, po.BHat = A0.copy(), B0.copy(); po.SigmaA = z.copy(); po.SigmaB = z.copy()
    po.CA = 0.1 * np.eye(4); po.CB = 0.1 * np.eye(4); po.invCA = 10 * np.eye(4); po.invCB = 10 * np.eye(4); po.sigma2 = 0.1
    Yf = Y.astype(np.float32).astype(np.float64)
    O.vbmf_(Yf, po, 1, eps=0.0, est_covs=True, est_var=True)
    assert relF(s1["AHat"], po.AHat) < 1e-4 and relF(s1["BHat"], po.BHat) < 1e-4


def test_two_handles_are_independent(pkg):
    """Distinct handles hold distinct problems; interleaved use does not leak state (one handle = one problem)."""
    rng = np.random.default_rng(8)
    probs = []
    for (L, M, H) in [(120, 80, 3), (90, 140, 6)]:
        Y = rng.standard_normal((L, M)).astype(np.float32).astype(np.float64)
        probs.append((Y, rng.standard_normal((M, H)), rng.standard_normal((L, H)), H))
    ctxs = [pkg.capi.Context(p[0].shape[0], p[0].shape[1], p[3], y_dtype=pkg.VBMF_Y_F32) for p in probs]
    try:
        for c, (Y, A0, B0, H) in zip(ctxs, probs):
            c.set_Y(Y)
            z = np.zeros((H, H))
            c.set_state(A0, B0, z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
        for _ in range(3):                       # interleave single sweeps
            for c in ctxs:
                c.run(1, eps=0.0, est_covs=True, est_var=True)
        inter = [c.get_state() for c in ctxs]
        for c, (Y, A0, B0, H) in zip(ctxs, probs):   # the same three sweeps, each handle alone
            z = np.zeros((H, H))
            c.set_state(A0, B0, z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
            c.run(3, eps=0.0, est_covs=True, est_var=True)
        alone = [c.get_state() for c in ctxs]
    finally:
        for c in ctxs:
            c.close()
    for a, b in zip(inter, alone):
        assert relF(a["AHat"], b["AHat"]) < 1e-6 and relF(a["BHat"], b["BHat"]) < 1e-6


def test_leading_dimension_and_views(pkg):
    """The boundary takes column-major data with a leading dimension (a Julia SubArray / a Fortran-ordered slice)."""
    rng = np.random.default_rng(21)
    big = np.asfortranarray(rng.standard_normal((100, 60)))
    Y = big[10:75, 5:45]                                   # 65 x 40 view, ld = 100
    with pkg.capi.Context(65, 40, 3, y_dtype=pkg.VBMF_Y_F32) as c:
        c.set_Y(Y)
        assert np.array_equal(c.get_Y(), Y.astype(np.float32).astype(np.float64))


def test_nan_in_Y_surfaces_as_an_error_or_nan_exit(pkg):
    """Julia propagates NaN (d becomes NaN and the loop exits); the library must not hang or return
    garbage silently: it stops after the first sweep with a NaN d or reports VBMF_ERR_NUMERIC."""
    rng = np.random.default_rng(2)
    Y = rng.standard_normal((40, 30)); Y[3, 4] = np.nan
    z = np.zeros((3, 3))
    with pkg.capi.Context(40, 30, 3, y_dtype=pkg.VBMF_Y_F32) as c:
        c.set_Y(Y)
        c.set_state(rng.standard_normal((30, 3)), rng.standard_normal((40, 3)), z, z, 0.1 * np.ones(3), 0.1 * np.ones(3), 0.1)
        try:
            it, d, _ = c.run(10, eps=1e-6, est_covs=True, est_var=True)
            assert it <= 2 and not (d > 1e-6)
        except pkg.VbmfError as e:
            assert e.code == -4


def test_argument_errors(pkg):
    capi = pkg.capi
    for bad in [(0, 5, 2), (5, 0, 2), (5, 5, 0), (-1, 5, 2)]:
        with pytest.raises(pkg.VbmfError) as ei:
            capi.Context(*bad)
        assert ei.value.code == -1
    with pytest.raises(pkg.VbmfError) as ei:
        capi.Context(10, 10, 257)
    assert ei.value.code == -6
    with pytest.raises(pkg.VbmfError):
        capi.Context(10, 10, 2, nranks=1, L_global=20)                  # L_global must equal L without sharding
    with capi.Context(10, 8, 2, y_dtype=pkg.VBMF_Y_F32) as c:
        with pytest.raises(pkg.VbmfError, match="no Y"):
            c.run(1)
        c.set_Y(np.ones((10, 8)))
        with pytest.raises(pkg.VbmfError, match="no state"):
            c.run(1)
        z = np.zeros((2, 2))
        with pytest.raises((pkg.VbmfError, ValueError)):
            c.set_state(np.ones((7, 2)), np.ones((10, 2)), z, z, np.ones(2), np.ones(2), 0.1)   # AHat has the wrong M
        c.set_state(np.ones((8, 2)), np.ones((10, 2)), z, z, np.ones(2), np.ones(2), 0.1)
        with pytest.raises(pkg.VbmfError):
            c.run(-1)
        with pytest.raises(pkg.VbmfError):
            c.step(1 << 9)
        with pytest.raises(pkg.VbmfError, match="sparse"):
            c.sparse_run(1)
        with pytest.raises(pkg.VbmfError):
            c.set_state(np.ones((8, 2)), np.ones((10, 2)), z, z, np.ones(2), np.ones(2), 0.1, labels0=[99], H1=1)  # label out of range
    with capi.Context(10, 8, 2, variant=capi.VBMF_VARIANT_SPARSE_DIAG) as c:
        with pytest.raises(pkg.VbmfError, match="sparse"):
            c.run(1)
