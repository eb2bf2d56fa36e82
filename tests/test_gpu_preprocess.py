"""Pre-processing fused into the upload: scaleY / preprocess of src/util.jl:36-86.
The device path (PreprocessPlan + set_Y_preprocessed) must hold exactly the oracle's lambda * scaleY(Y)[kept rows],
rounded once to the device dtype.  PARITY UNPINNED beyond the oracle (the reference records no pre-processed matrix)."""
import numpy as np
import pytest

import __graft_entry__ as G
from oracle import vbmf_oracle as O
from tests.helpers import bf16_round, relF, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    G.build()
    return G.load_package()


def _raw_matrix(L, M, seed):
    """Feature-matrix-like input: rows with very different offsets and scales, plus the degenerate rows the
    reference's filter exists for: constant rows, all-zero rows, and a row that varies only below the 1e-8 floor."""
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((L, M)) * rng.uniform(0.01, 50.0, (L, 1)) + rng.uniform(-100.0, 100.0, (L, 1))
    Y[3] = 7.25
    Y[11] = 0.0
    Y[17] = 1.0 + 1e-10 * rng.standard_normal(M)
    Y[L - 1] = -3.0
    return Y


@pytest.mark.parametrize("ydt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(64, 40), (301, 157), (1000, 333)])
def test_preprocess_fused_upload(pkg, ydt, shape):
    L, M = shape
    Y = _raw_matrix(L, M, 31 + L)
    lam = 0.7
    ref, used = O.preprocess(Y, lam, return_rows=True)
    assert used.size == L - 4 and not np.isin([3, 11, 17, L - 1], used).any()
    # host mirror with the reference's call shape
    assert np.array_equal(pkg.preprocess(Y, lam), ref)
    dt = pkg.VBMF_Y_F32 if ydt == "f32" else pkg.VBMF_Y_BF16
    s, rows = pkg.preprocessed_session(Y, lam, 4, y_dtype=dt, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    try:
        assert np.array_equal(rows, used) and s.L == used.size
        got = s.ctx.get_Y()
        want = ref.astype(np.float32).astype(np.float64) if ydt == "f32" else bf16_round(ref)
        # row statistics are summed in another order than numpy's pairwise sums: a last-place difference in mu/den may
        # move an entry across a rounding boundary of the device dtype
        ulp = 2.0 ** -23 if ydt == "f32" else 2.0 ** -8
        bad = np.abs(got - want) > 1.01 * ulp * np.maximum(np.abs(want), 1e-30)
        report(f"preprocess fused upload {L}x{M} {ydt}: kept {used.size}/{L} rows, entries off by one ulp: "
               f"{int(np.count_nonzero(got != want))}, beyond: {int(bad.sum())}, relF {relF(got, want):.2e}")
        assert not bad.any()
        assert np.count_nonzero(got != want) <= 1e-3 * got.size
        assert abs(s.ctx.trYY() - float(np.sum(got * got))) <= 1e-12 * s.ctx.trYY()
    finally:
        s.close()


def test_preprocess_then_factorize(pkg):
    """The reference's pipeline (examples/mil_util.jl:829 then vbmf!): pre-process on the device, factorize the result."""
    L, M, H = 400, 260, 5
    rng = np.random.default_rng(12)
    Bs = rng.standard_normal((L, H)) * np.linspace(1.0, 2.5, H)
    As = np.zeros((M, H)); As[np.arange(M), rng.integers(0, H, M)] = 1.0
    Y = (Bs @ As.T + 0.05 * rng.standard_normal((L, M))) * rng.uniform(0.5, 20.0, (L, 1)) + rng.uniform(-5, 5, (L, 1))
    Y[7] = 2.0
    s, rows = pkg.preprocessed_session(Y, 1.0, H, y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    try:
        Ys = np.ascontiguousarray(s.ctx.get_Y())
        po = O.vbmf_init(Ys, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(3), materialize_yhat=False)
        s.ctx.set_state(po.AHat, po.BHat, po.SigmaA, po.SigmaB, np.diag(po.CA), np.diag(po.CB), po.sigma2)
        it, d, _ = s.run(5, eps=0.0, est_covs=True, est_var=True)
        st = s.ctx.get_state()
        O.vbmf_(Ys, po, 5, eps=0.0, est_covs=True, est_var=True)
        assert rows.size == L - 1 and it == 5
        assert relF(st["AHat"], po.AHat) < 2e-4 and relF(st["BHat"], po.BHat) < 2e-4
        assert abs(st["sigma2"] - po.sigma2) < 2e-3 * po.sigma2
    finally:
        s.close()
