"""lambda_max of an H x H Gram -- the spectral norm behind `delta` (src/util.jl:27-29; Julia 0.5's norm(::Matrix) is the largest singular
value) -- by the device kernels the run loop uses (H <= 64: repeated squaring + Rayleigh quotient; H > 64: Lanczos on the register-resident
matrix, ctrl_kernels.hpp), against numpy.linalg.eigvalsh on spectra chosen to be hard: a FLAT top cluster (what the delta-Gram of an over-ranked
fit looks like: the power iteration this replaced stopped at ~1e-4 there after 2048 steps), near-degenerate pairs, rank deficiency, the identity
(invariant subspace after one step), a dominant eigenvalue."""
import numpy as np
import pytest

import __graft_entry__ as G
from tests.helpers import report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    G.build()
    return G.load_package()


def _with_spectrum(H, lam, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((H, H)))
    return (Q * lam) @ Q.T


def _spectra(H, rng):
    flat = 1.0 + 1e-3 * rng.random(H)                                    # everything within 0.1 %
    mp = np.linalg.eigvalsh(np.cov(rng.standard_normal((H, 3 * H))))     # a Marchenko-Pastur bulk: no gap at the top edge
    pair = np.concatenate([[1.0, 1.0 - 1e-5], 0.5 * rng.random(H - 2)])  # a near-degenerate leading pair
    lowrank = np.concatenate([np.linspace(3.0, 1.0, 7), np.zeros(H - 7)])
    dominant = np.concatenate([[50.0], rng.random(H - 1)])
    decay = 2.0 ** -np.arange(H, dtype=np.float64)                       # 2^-k: most of it far below the trace's rounding
    return dict(flat=flat, marchenko_pastur=mp, pair=pair, lowrank=lowrank, dominant=dominant, decay=decay, identity=np.ones(H))


@pytest.mark.parametrize("H", [40, 128, 130, 200, 256])
def test_lambda_max_on_hard_spectra(pkg, H):
    capi = pkg.capi
    rng = np.random.default_rng(1000 + H)
    with capi.Context(600, 300, H, y_dtype=capi.VBMF_Y_F32) as c:
        worst = 0.0
        for name, lam in _spectra(H, rng).items():
            Gm = _with_spectrum(H, lam, 7 + H)
            Gm = 0.5 * (Gm + Gm.T)
            ref = float(np.linalg.eigvalsh(Gm)[-1])
            got, us = c.lambda_max(Gm)
            err = abs(got - ref) / ref
            worst = max(worst, err)
            report(f"lambda_max H={H} {name}: lam={err:.2e}  [{us:.0f} us]")
            # H > 64 (Lanczos), measured: <= 8e-8 on every spectrum.  H <= 64 keeps the repeated squaring (faster inside the short pass
            # launches of narrow problems, ctrl_kernels.hpp): exact off clusters (<= 1e-7 measured), inside one its documented bound is
            # n / (2e 2^11) -- measured 2.2e-4 on the flat spectrum; `delta` takes the square root: half of it
            tol = 2e-6 if H > 64 else (5e-4 if name in ("flat", "marchenko_pastur", "pair") else 2e-6)
            assert err <= tol, (H, name, got, ref, err)


@pytest.mark.parametrize("H", [12, 40, 64])
def test_lambda_max_exact_mode_at_small_ranks(pkg, H):
    """VBMF_DEBUG_EXACT_LAMBDA (environment VBMF_EXACT_LAMBDA=1): the Lanczos iteration at H <= 64 too -- the flat spectrum that the default
    squaring resolves to 2.2e-4 comes out to <= 1e-7 (measured 6e-8), like every other one."""
    capi = pkg.capi
    rng = np.random.default_rng(2000 + H)
    with capi.Context(600, 300, H, y_dtype=capi.VBMF_Y_F32) as c:
        c.debug_set(capi.DEBUG_EXACT_LAMBDA, 1)
        for name, lam in _spectra(H, rng).items():
            Gm = _with_spectrum(H, lam, 9 + H)
            Gm = 0.5 * (Gm + Gm.T)
            ref = float(np.linalg.eigvalsh(Gm)[-1])
            got, us = c.lambda_max(Gm)
            err = abs(got - ref) / ref
            report(f"lambda_max exact mode H={H} {name}: lam={err:.2e}  [{us:.0f} us]")
            assert err <= 2e-6, (H, name, got, ref, err)


def test_exact_mode_run_matches_default_where_the_spectrum_is_separated(pkg):
    """A whole run with the switch on: same stopping sweep and state as the default on a problem whose Grams have separated spectra (the two
    methods agree to 1e-7 there), so the switch changes nothing but the clustered cases."""
    from tests.helpers import to_pkg_params
    from oracle import vbmf_oracle as O
    import os
    L, M, H = 300, 170, 5
    rng = np.random.default_rng(77)
    Y = rng.standard_normal((L, H)) @ (rng.standard_normal((H, M)) * np.linspace(3.0, 1.0, H)[:, None]) + 0.1 * rng.standard_normal((L, M))
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    po = O.vbmf_init(Y, H, ca=1.0, cb=1.0, sigma2=1.0, rng=np.random.default_rng(5), materialize_yhat=False)
    res = []
    for exact in ("0", "1"):
        os.environ["VBMF_EXACT_LAMBDA"] = exact
        try:
            pkg.invalidate()
            pg = to_pkg_params(pkg, po)
            pkg.vbmf_(Y, pg, 60, eps=1e-4, est_covs=True, est_var=True)
            res.append((pg._last_run[0], pg._last_run[1], pg.AHat.copy()))
        finally:
            os.environ.pop("VBMF_EXACT_LAMBDA", None)
            pkg.invalidate()
    assert res[0][0] == res[1][0]
    assert abs(res[0][1] - res[1][1]) <= 1e-6 * res[0][1]
    assert np.array_equal(res[0][2], res[1][2])


def test_lambda_max_degenerate_inputs(pkg):
    capi = pkg.capi
    H = 160
    with capi.Context(600, 300, H, y_dtype=capi.VBMF_Y_F32) as c:
        got, _ = c.lambda_max(np.zeros((H, H)))
        assert got == 0.0                                                  # tr = 0: nothing to iterate on
        e = np.zeros((H, H)); e[17, 17] = 2.5                              # rank one, the start vector is not orthogonal to it
        got, _ = c.lambda_max(e)
        assert abs(got - 2.5) <= 1e-6 * 2.5
        v = np.random.default_rng(3).standard_normal(H)
        got, _ = c.lambda_max(np.outer(v, v))
        assert abs(got - v @ v) <= 1e-6 * (v @ v)
