"""Fixed-basis inference: vbls! + copy_vbmf_params of examples/mil_util.jl:179-236 --
the device loop (vbmf_run_fixed_basis / vbmf_sparse_run_fixed_basis: Y'B formed once, B frozen) against the oracle's
restatement, which repeats the reference's full updates.  PARITY UNPINNED (no recorded vbls! run in the reference)."""
import numpy as np
import pytest

import __graft_entry__ as G
from oracle import vbmf_oracle as O
from tests.helpers import compare, relF, report, to_pkg_params
from tests.test_gpu_sparse import _cmp, _to_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    G.build()
    return G.load_package()


def _train_and_new_bag(L, M, M2, H, seed):
    """A factorization trained on Y (L x M), then a NEW matrix with the same row space and another M (a 'bag')."""
    rng = np.random.default_rng(seed)
    Bs = rng.standard_normal((L, H)) * np.linspace(1.0, 2.5, H)
    def draw(m):
        As = np.zeros((m, H)); As[np.arange(m), rng.integers(0, H, m)] = 1.0
        return Bs @ As.T + 0.05 * rng.standard_normal((L, m))
    return draw(M), draw(M2)


@pytest.mark.parametrize("mode", ["f32", "bf16x2"])
def test_vbls_basic_device_loop(pkg, mode):
    L, M, M2, H = 500, 300, 177, 6
    Y, Y2 = _train_and_new_bag(L, M, M2, H, 77)
    ydt = pkg.VBMF_Y_F32 if mode == "f32" else pkg.VBMF_Y_BF16
    pkg.set_defaults(y_dtype=ydt, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    with pkg.capi.Context(L, M2, H, y_dtype=ydt) as c:       # the bag exactly as the device stores it
        c.set_Y(Y2)
        Y2s = np.ascontiguousarray(c.get_Y())
    po = O.vbmf_init(Y, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(5), materialize_yhat=False)
    O.vbmf_(Y, po, 30, eps=0.0, est_covs=True, est_var=True)
    qo = O.copy_vbmf_params(Y2s, po, rng=np.random.default_rng(6))
    qg = pkg.copy_vbmf_params(Y2s, to_pkg_params(pkg, po), rng=np.random.default_rng(6))
    assert np.array_equal(qg.AHat, qo.AHat) and np.array_equal(qg.BHat, po.BHat) and qg.M == M2
    A_gpu = pkg.vbls_(Y2s, qg, 20)
    O.vbls_(Y2s, qo, 20)
    # B round-trips through the device's factor storage (fp32; hi+lo bf16 = 2^-17 relative in the bf16 modes)
    assert A_gpu is qg.AHat and relF(qg.BHat, po.BHat) < 1e-5
    tol = dict(default=2e-4, sigma2=2e-3) if mode == "f32" else dict(default=4e-4, sigma2=4e-3)
    compare(f"vbls! device loop, 20 iterations, {mode}", qg, qo, tol, fields=("AHat", "SigmaA", "CA"))
    assert relF(qg.YHat, qo.BHat @ qo.AHat.T) < 1e-3


@pytest.mark.parametrize("H,niter", [(3, 150), (20, 40), (40, 25), (64, 12)])
def test_vbls_hxh_loop_against_the_general_path_and_the_oracle(pkg, monkeypatch, H, niter):
    """Without a label mask vbls! is H x H algebra on S = (Y'B)'(Y'B) (ctrl_kernels.hpp, vbls_loop_kernel): all but the first
    iteration run in one launch.  Same end state as the general path (VBMF_VBLS_LOOP=0: every iteration with its own post /
    Gram / dot kernels) and as the oracle's literal loop -- one, two and four 16-blocks per side."""
    L, M, M2 = 700, 500, 230
    Y, Y2 = _train_and_new_bag(L, M, M2, H, 400 + H)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Y2s = Y2.astype(np.float32).astype(np.float64)
    po = O.vbmf_init(Y, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(15), materialize_yhat=False)
    O.vbmf_(Y, po, 20, eps=0.0, est_covs=True, est_var=True)
    qo = O.copy_vbmf_params(Y2s, po, rng=np.random.default_rng(16))
    O.vbls_(Y2s, qo, niter)
    out = {}
    for name, env in (("loop", "1"), ("general", "0")):
        monkeypatch.setenv("VBMF_VBLS_LOOP", env)
        qg = pkg.copy_vbmf_params(Y2s, to_pkg_params(pkg, po), rng=np.random.default_rng(16))
        pkg.vbls_(Y2s, qg, niter)
        out[name] = qg
        # sigma2 is the reference's own cancellation (||Y||^2 - 2 tr + tr: signal / noise ~ 400) of fp32-stored quantities, and
        # SigmaA = sigma2 inv(K) inherits it: measured 5.3e-4 after 150 iterations at H = 3; the factor itself 2.6e-6
        compare(f"vbls! {name} path H{H} x{niter}", qg, qo, dict(default=3e-4, SigmaA=2e-3, sigma2=2e-3), fields=("AHat", "SigmaA", "CA"))
    a, b = out["loop"], out["general"]
    assert relF(a.AHat, b.AHat) < 2e-5 and relF(a.SigmaA, b.SigmaA) < 2e-3 and abs(a.sigma2 - b.sigma2) < 2e-3 * b.sigma2


def test_sessions_of_dead_matrices_are_reused(pkg, monkeypatch):
    """A caller that walks over many small matrices of one shape (the MIL classifier's bags) must not pay a context creation per
    matrix: once a matrix is garbage, its session serves the next matrix of the same shape, rank and storage defaults -- with
    the same results as a fresh session."""
    import gc
    L, M, M2, H = 230, 300, 60, 10
    Y, _ = _train_and_new_bag(L, M, M2, H, 55)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    po = O.vbmf_init(Y, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(25), materialize_yhat=False)
    O.vbmf_(Y, po, 10, eps=0.0, est_covs=True, est_var=True)
    rng = np.random.default_rng(3)
    bags = [rng.standard_normal((L, M2)) for _ in range(4)]
    pkg.invalidate()
    created = []
    real = pkg.Session.__init__

    def counting(self, *a, **k):
        created.append(a[:3])
        real(self, *a, **k)
    monkeypatch.setattr(pkg.Session, "__init__", counting)
    outs = []
    for b in bags:
        Yb = b.copy()
        qg = pkg.copy_vbmf_params(Yb, to_pkg_params(pkg, po), rng=np.random.default_rng(26))
        pkg.vbls_(Yb, qg, 12)
        outs.append(qg.AHat.copy())
        del Yb, qg
        gc.collect()
    assert len(created) == 1, created                            # one context for the four bags
    pkg.invalidate()
    for b, a in zip(bags, outs):                                 # ... and the same numbers as a fresh session per bag
        qg = pkg.copy_vbmf_params(b, to_pkg_params(pkg, po), rng=np.random.default_rng(26))
        pkg.vbls_(b, qg, 12)
        assert np.array_equal(qg.AHat, a)
        pkg.invalidate()


def test_vbls_sparse_device_loop(pkg):
    L, M, M2, H = 400, 260, 150, 5
    Y, Y2 = _train_and_new_bag(L, M, M2, H, 91)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Y2s = Y2.astype(np.float32).astype(np.float64)
    po = O.vbmf_sparse_init(Y, H, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(7), full_cov=False,
                            materialize_yhat=False)
    O.vbmf_sparse_(Y, po, 12, eps=0.0, full_cov=False)
    qo = O.copy_vbmf_params(Y2s, po, rng=np.random.default_rng(8))
    qg = pkg.copy_vbmf_params(Y2s, _to_pkg(pkg, po), rng=np.random.default_rng(8))
    assert np.array_equal(qg.ATVecHat, qo.ATVecHat) and qg.gamma == po.gamma
    pkg.vbls_(Y2s, qg, 10)
    O.vbls_sparse_(Y2s, qo, 10)
    _cmp("vbls! sparse device loop, 10 iterations", qg, qo, 2e-3,
         fields=("ATVecHat", "diagSigmaATVec", "SigmaA", "CA", "beta"))
    assert relF(qg.BHat, po.BHat) < 1e-6
