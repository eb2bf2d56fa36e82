"""GPU parity of the ARD-sparse variant (src/vbmf_sparse.jl, full_cov=false, diag_var=false) against the
oracle's diagonal branch.  PARITY UNPINNED: the reference recorded only a full_cov=true run; the diagonal
branch (incl. the QS1 `repeat(v, inner=M-1)` layout and QS2) and lowerBound follow from source reading and
are checked against the oracle alone (the oracle's shared parts are pinned by the sparse fixture)."""
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


def _mk(L, M, H, seed, **kw):
    rng = np.random.default_rng(seed)
    Y, A, B = O.toy_matrix(L, M, H, 0.05, rng)
    Y = (B * np.linspace(1.0, 2.5, H)) @ A.T + 0.05 * rng.standard_normal((L, M))
    po = O.vbmf_sparse_init(Y, H, ca=0.1, cb=0.1, sigma=0.1, rng=np.random.default_rng(seed + 1), full_cov=False,
                            materialize_yhat=False, **kw)
    return Y, po


def _to_pkg(pkg, po):
    p = pkg.vbmf_sparse_parameters()
    for f in ("L", "M", "H", "MH", "H1", "alpha0", "beta0", "alpha", "gamma0", "delta0", "gamma", "sigmaHat", "eta0", "zeta0",
              "eta", "zeta", "trYTY"):
        setattr(p, f, getattr(po, f))
    p.labels = np.asarray(po.labels, dtype=np.int64) + 1
    for f in ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta"):
        setattr(p, f, getattr(po, f).copy())
    return p


FIELDS = ("ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta")


def _cmp(tag, pg, po, tol, fields=FIELDS):
    errs = {f: relF(getattr(pg, f), getattr(po, f)) for f in fields}
    errs["sigmaHat"] = abs(pg.sigmaHat - po.sigmaHat) / po.sigmaHat
    report(f"sparse {tag}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, (tag, bad)


@pytest.mark.parametrize("L,M,H", [(10, 20, 2), (300, 170, 5), (500, 260, 40)])
def test_sparse_each_update_f32(pkg, L, M, H):
    Y, po = _mk(L, M, H, 40 + H)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    for sweep in range(3):
        pg = _to_pkg(pkg, po)
        pkg.sparse_updateA_(Yf, pg); O.sparse_updateA(Yf, po, full_cov=False)
        _cmp(f"{L}x{M} H{H} s{sweep} updateA", pg, po, 5e-5, ("ATVecHat", "diagSigmaATVec", "SigmaA"))
        pg = _to_pkg(pkg, po)
        pkg.sparse_updateB_(Yf, pg); O.sparse_updateB(Yf, po)
        _cmp(f"{L}x{M} H{H} s{sweep} updateB", pg, po, 5e-5, ("BHat", "SigmaB"))
        pg = _to_pkg(pkg, po)
        pkg.sparse_updateCA_(pg, Y=Yf); pkg.sparse_updateCB_(pg, Y=Yf); O.sparse_updateCA(po); O.sparse_updateCB(po)
        _cmp(f"{L}x{M} H{H} s{sweep} updateC", pg, po, 5e-5, ("CA", "beta", "CB", "delta"))
        pg = _to_pkg(pkg, po)
        pkg.sparse_updateSigma_(Yf, pg); O.sparse_updateSigma(Yf, po)
        _cmp(f"{L}x{M} H{H} s{sweep} updateSigma", pg, po, 5e-4, ())
        assert abs(pg.zeta - po.zeta) / po.zeta < 5e-4


@pytest.mark.parametrize("compat", [True, False])
def test_sparse_repeat_layout_quirk(pkg, compat):
    """QS1: reference_compat reproduces repeat(v, inner=M-1); switching it off gives the consistent tiling."""
    L, M, H = 200, 90, 4
    Y, po = _mk(L, M, H, 77)
    Yf = Y.astype(np.float32).astype(np.float64)
    rc = pkg.capi.VBMF_COMPAT_DEFAULT if compat else (pkg.capi.VBMF_COMPAT_DEFAULT & ~pkg.capi.VBMF_COMPAT_SPARSE_REPEAT)
    hyper = dict(alpha0=po.alpha0, beta0=po.beta0, gamma0=po.gamma0, delta0=po.delta0, eta0=po.eta0, zeta0=po.zeta0)
    # make B's columns have clearly different norms so the two layouts differ
    po.BHat = po.BHat * np.array([1.0, 2.0, 3.0, 4.0])
    with pkg.capi.Context(L, M, H, y_dtype=pkg.VBMF_Y_F32, variant=pkg.capi.VBMF_VARIANT_SPARSE_DIAG, reference_compat=rc) as c:
        c.set_Y(Yf)
        c.sparse_set_state(po.ATVecHat, po.diagSigmaATVec, po.CA, po.beta, po.BHat, po.SigmaB, po.CB, po.delta, po.sigmaHat,
                           po.zeta, hyper)
        c.sparse_step(pkg.SSTEP_A)
        s = c.sparse_get_state()
    O.sparse_updateA(Yf, po, full_cov=False, reference_compat=compat)
    assert relF(s["diagSigmaATVec"], po.diagSigmaATVec) < 5e-6
    assert relF(s["ATVecHat"], po.ATVecHat) < 5e-5
    other = O.spread_v(np.arange(1.0, H + 1), M, not compat)
    assert not np.array_equal(other, O.spread_v(np.arange(1.0, H + 1), M, compat))


@pytest.mark.parametrize("mode", ["f32", "bf16x2"])
def test_sparse_run_and_lower_bound(pkg, mode):
    L, M, H = 600, 380, 6
    Y, po = _mk(L, M, H, 21, H1=2, labels=[3, 50, 200])
    ydt = pkg.VBMF_Y_F32 if mode == "f32" else pkg.VBMF_Y_BF16
    with pkg.capi.Context(L, M, H, y_dtype=ydt) as c:
        c.set_Y(Y)
        Ys = np.ascontiguousarray(c.get_Y())
    po.trYTY = float(np.sum(Ys * Ys))
    pkg.set_defaults(y_dtype=ydt, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg(pkg, po)
    d_gpu = pkg.vbmf_sparse_(Ys, pg, 15, eps=0.0, est_cb=True)
    d_ref, n = O.vbmf_sparse_(Ys, po, 15, eps=0.0, full_cov=False, est_cb=True)
    tol = 2e-3 if mode == "f32" else 5e-3
    _cmp(f"run15 {mode}", pg, po, tol)
    assert pg._last_run[0] == 15 and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6
    assert np.all(pg.AHat[[3, 50, 200], H - 2:] == 0.0) and np.all(pg.AHat[[2, 49, 199], H - 2:] != 0.0)
    lb_gpu = pkg.lowerBound(Ys, pg)
    lb_ref = O.lowerBound(Ys, po)
    report(f"sparse lowerBound {mode}: gpu {lb_gpu:.6f} oracle {lb_ref:.6f}")
    assert abs(lb_gpu - lb_ref) <= 2e-3 * abs(lb_ref)
    # lowerBound of the ORACLE's state evaluated on the device: isolates the bound itself from trajectory drift
    pg2 = _to_pkg(pkg, po)
    lb2 = pkg.lowerBound(Ys, pg2)
    assert abs(lb2 - lb_ref) <= 1e-5 * abs(lb_ref) + 1e-3, (lb2, lb_ref)
    lb3 = pkg.lowerBound(Ys, pg2, clamp=False)
    assert abs(lb3 - O.lowerBound(Ys, po, clamp=False)) <= 1e-5 * abs(lb_ref) + 1e-3


def test_sparse_termination(pkg, golden_dir):
    """Loop test on the device (src/vbmf_sparse.jl:368): the reference's own toy matrix (the Y of its recorded
    sparse run), diagonal branch, eps = 1e-3."""
    import os
    g = np.load(os.path.join(golden_dir, "sparse_test.npz"))
    Y = g["Y"]
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32)
    Yf = Y.astype(np.float32).astype(np.float64)
    # (with ca=cb=sigma=0.1 this model collapses to B = 0 and the reference's loop exits on d = NaN after ~43
    # sweeps; the default hyper-parameters 1, 1, 1 converge normally)
    po = O.vbmf_sparse_init(Yf, 2, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(12), full_cov=False,
                            materialize_yhat=False)
    pg = _to_pkg(pkg, po)
    d_ref, n = O.vbmf_sparse_(Yf, po, 400, eps=1e-3, full_cov=False)
    d_gpu = pkg.vbmf_sparse_(Yf, pg, 400, eps=1e-3)
    report(f"sparse termination: oracle n={n} d={d_ref:.3e}; gpu n={pg._last_run[0]} d={d_gpu:.3e}")
    assert 3 < n < 400 and abs(pg._last_run[0] - n) <= 2 and d_gpu <= 1e-3
    _cmp("termination", pg, po, 5e-3)


# ---- heteroscedastic rows: diag_var = true ------------------------------------
def _mk_hetero(L, M, H, seed, **kw):
    """toy data whose rows have different noise levels (what diag_var models)."""
    rng = np.random.default_rng(seed)
    Y, A, B = O.toy_matrix(L, M, H, 0.0, rng)
    Y = (B * np.linspace(1.0, 2.5, H)) @ A.T + rng.uniform(0.02, 0.4, (L, 1)) * rng.standard_normal((L, M))
    po = O.vbmf_sparse_init(Y, H, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(seed + 1), full_cov=False,
                            materialize_yhat=False, **kw)
    return Y, po


def _to_pkg_hetero(pkg, po):
    p = _to_pkg(pkg, po)
    p.sigmaVecHat, p.etaVec, p.zetaVec = po.sigmaVecHat.copy(), po.etaVec.copy(), po.zetaVec.copy()
    return p


HFIELDS = FIELDS + ("sigmaVecHat", "zetaVec")


def _cmp_h(tag, pg, po, tol):
    errs = {f: relF(getattr(pg, f), getattr(po, f)) for f in HFIELDS}
    report(f"sparse diag_var {tag}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, (tag, bad)


@pytest.mark.parametrize("L,M,H", [(10, 20, 2), (300, 170, 5), (500, 260, 40)])
def test_hetero_each_update_f32(pkg, L, M, H):
    """updateA!/updateB!/updateCA!/updateCB!/updateSigma! with diag_var=true (src/vbmf_sparse.jl:207-212,229-230,256-261,
    308-315), each from the oracle's state after a few heteroscedastic sweeps.  PARITY UNPINNED (no recorded run)."""
    Y, po = _mk_hetero(L, M, H, 300 + L)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    O.vbmf_sparse_(Yf, po, 3, eps=0.0, full_cov=False, diag_var=True)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    import copy
    for name, fo, fg in [("A", lambda q: O.sparse_updateA(Yf, q, diag_var=True), lambda q: pkg.sparse_updateA_(Yf, q, diag_var=True)),
                         ("B", lambda q: O.sparse_updateB(Yf, q, diag_var=True), lambda q: pkg.sparse_updateB_(Yf, q, diag_var=True)),
                         ("Sigma", lambda q: O.sparse_updateSigma(Yf, q, diag_var=True), lambda q: pkg.sparse_updateSigma_(Yf, q, diag_var=True))]:
        qo = copy.deepcopy(po)
        qg = _to_pkg_hetero(pkg, po)
        fo(qo); fg(qg)
        _cmp_h(f"update {name} {L}x{M} H={H}", qg, qo, 2e-5)


@pytest.mark.parametrize("L,M,H", [(900, 500, 128), (700, 420, 160)])
def test_hetero_run_wide_rank(pkg, L, M, H):
    """diag_var = true INSIDE the run loop at H >= 128 (16 / 32 accumulator tiles per wave, un-fused post and Gram kernels): the row
    scaling of B happens after the post kernel, from the fp32 factor, so that kernel must keep storing it and must not leave
    delta tiles of the un-scaled product (it did neither in the run loops before this test existed)."""
    Y, po = _mk_hetero(L, M, H, 78)
    with pkg.capi.Context(L, M, H, y_dtype=pkg.VBMF_Y_BF16) as c:
        c.set_Y(Y)
        Ys = np.ascontiguousarray(c.get_Y())
    po.trYTY = float(np.sum(Ys * Ys))
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_BF16, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg_hetero(pkg, po)
    d_gpu = pkg.vbmf_sparse_(Ys, pg, 4, eps=0.0, diag_var=True)
    d_ref, n = O.vbmf_sparse_(Ys, po, 4, eps=0.0, full_cov=False, diag_var=True)
    _cmp_h(f"run4 bf16x2 {L}x{M} H{H}", pg, po, 5e-3)
    assert pg._last_run[0] == 4 and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6


@pytest.mark.parametrize("mode", ["f32", "bf16x2"])
def test_hetero_run(pkg, mode):
    L, M, H = 600, 380, 6
    Y, po = _mk_hetero(L, M, H, 77, H1=2, labels=[3, 50, 200])
    ydt = pkg.VBMF_Y_F32 if mode == "f32" else pkg.VBMF_Y_BF16
    with pkg.capi.Context(L, M, H, y_dtype=ydt) as c:
        c.set_Y(Y)
        Ys = np.ascontiguousarray(c.get_Y())
    po.trYTY = float(np.sum(Ys * Ys))
    pkg.set_defaults(y_dtype=ydt, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    pg = _to_pkg_hetero(pkg, po)
    d_gpu = pkg.vbmf_sparse_(Ys, pg, 8, eps=0.0, diag_var=True)
    d_ref, n = O.vbmf_sparse_(Ys, po, 8, eps=0.0, full_cov=False, diag_var=True)
    _cmp_h(f"run8 {mode}", pg, po, 2e-3 if mode == "f32" else 5e-3)
    assert pg._last_run[0] == 8 and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6
    # the rows' precisions track the rows' noise levels (what the model is for)
    assert np.corrcoef(np.log(pg.sigmaVecHat), np.log(po.sigmaVecHat))[0, 1] > 0.999
    # fixed-basis inference with heteroscedastic rows (vbls!, diag_var = true): both passes every iteration
    qo = O.copy_vbmf_params(Ys, po, rng=np.random.default_rng(4))
    qg = pkg.copy_vbmf_params(Ys, pg, rng=np.random.default_rng(4))
    qo.BHat, qo.SigmaB, qo.CB = po.BHat.copy(), po.SigmaB.copy(), po.CB.copy()
    qg.BHat, qg.SigmaB, qg.CB = po.BHat.copy(), po.SigmaB.copy(), po.CB.copy()
    pkg.vbls_(Ys, qg, 5, diag_var=True)
    O.vbls_sparse_(Ys, qo, 5, diag_var=True)
    errs = {f: relF(getattr(qg, f), getattr(qo, f)) for f in ("ATVecHat", "diagSigmaATVec", "CA", "sigmaVecHat")}
    report(f"sparse diag_var vbls5 {mode}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert max(errs.values()) < (2e-3 if mode == "f32" else 5e-3), errs


# ---- full_cov = true (src/vbmf_sparse.jl:178-202): the branch the reference's recorded sparse run used ------------------
def _fixture_params(pkg, g, t):
    p = pkg.vbmf_sparse_parameters()
    p.L, p.M, p.H, p.H1 = int(g["L"][t]), int(g["M"][t]), int(g["H"][t]), int(g["H1"][t])
    p.MH = p.M * p.H
    for f in ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta"):
        setattr(p, f, g[f][t].copy())
    for f in ("alpha0", "beta0", "alpha", "gamma0", "delta0", "gamma", "sigmaHat", "eta0", "zeta0", "eta", "zeta", "trYTY"):
        setattr(p, f, float(g[f][t]))
    return p


RFIELDS = ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta")


def test_full_cov_each_update_against_the_reference_record(pkg, golden_dir):
    """PINNED: every recorded slice t of the reference's own sparse run (examples/data/sparse_test/log.jld, full_cov = true,
    L = 10, M = 20, H = 2) is loaded on the device, ONE sweep is run there, and the result is compared with recorded slice
    t + 1 -- updateA! full branch (:178-202; the device inverts the M diagonal H x H blocks of the reference's dense
    40 x 40 matrix), updateB!, updateCA!, updateCB!, updateSigma!.  fp32 Y and fp32-stored factors: 1e-5."""
    import os
    g = np.load(os.path.join(golden_dir, "sparse_test.npz"))
    Y = np.ascontiguousarray(g["Y"])
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    worst = {}
    for t in range(0, 100):
        p = _fixture_params(pkg, g, t)
        pkg.vbmf_sparse_(Y, p, 1, eps=0.0, full_cov=True, est_cb=True)
        for f in RFIELDS:
            worst[f] = max(worst.get(f, 0.0), relF(getattr(p, f), g[f][t + 1]))
        for f in ("sigmaHat", "zeta"):
            worst[f] = max(worst.get(f, 0.0), abs(getattr(p, f) - g[f][t + 1]) / abs(g[f][t + 1]))
    report("sparse full_cov, one device sweep from each of the reference's 100 recorded slices, worst: "
           + " ".join(f"{k}={v:.2e}" for k, v in worst.items()))
    assert max(worst.values()) < 1e-5, worst
    # the blocks the device inverted are the diagonal blocks of the recorded dense covariance
    cov = {int(s): i for i, s in enumerate(g["cov_slices"])}
    t = max(s for s in cov if s > 0)
    p = _fixture_params(pkg, g, t - 1)
    pkg.sparse_updateA_(Y, p, full_cov=True)
    S = g["SigmaATVec"][cov[t]]
    H, M = p.H, p.M
    assert relF(p.diagSigmaATVec, np.diag(S)) < 1e-5
    assert relF(p.SigmaA, sum(S[m * H:(m + 1) * H, m * H:(m + 1) * H] for m in range(M))) < 1e-5
    off = S.copy()
    for m in range(M):
        off[m * H:(m + 1) * H, m * H:(m + 1) * H] = 0.0
    assert np.all(off == 0.0)                                   # the recorded matrix IS block diagonal


def test_full_cov_logged_trajectory_against_the_reference_record(pkg, golden_dir, tmp_path):
    """PINNED: the reference's recorded sparse experiment run on the device from slice 0 with per-sweep logging
    (vbmf_sparse!(...; full_cov = true, logdir = ...)); every one of the 101 slices is compared with the reference's log."""
    import os
    g = np.load(os.path.join(golden_dir, "sparse_test.npz"))
    Y = np.ascontiguousarray(g["Y"])
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    p = _fixture_params(pkg, g, 0)
    d = pkg.vbmf_sparse_(Y, p, 100, eps=0.0, full_cov=True, est_cb=True, logdir=str(tmp_path), desc="sparse_fixture")
    log, Yl, _ = pkg.load_log(os.path.join(str(tmp_path), "sparse_fixture"))
    assert p._last_run[0] == 100 and np.array_equal(Yl, Y) and log["AHat"].shape == (20, 2, 101)
    worst = {}
    for f in RFIELDS:
        worst[f] = max(relF(log[f][..., t], g[f][t]) for t in range(101))
    worst["sigmaHat"] = float(np.max(np.abs(log["sigmaHat"] - g["sigmaHat"]) / g["sigmaHat"]))
    report("sparse full_cov logged trajectory vs the reference's recorded log, worst slice: "
           + " ".join(f"{k}={v:.2e}" for k, v in worst.items()))
    assert max(worst.values()) < 2e-3, worst
    assert abs(p.sigmaHat - 4.60278260971759) < 2e-3 * 4.6           # SURVEY Appendix B known-answer values
    assert abs(p.zeta - 21.72598805535064) < 2e-3 * 21.7


# (60 000 x 26, H = 128: the shape of the round-2 advisor's finding -- 1 876 row tiles, where the weighted Gram's chunking wrote 118
#  slabs into a buffer sized for 94; the buffer is sized for every chunking now and every Gram launcher checks its slab count)
@pytest.mark.parametrize("L,M,H", [(300, 150, 6), (420, 60, 40), (400, 30, 100), (60000, 26, 128)])
def test_full_cov_with_heteroscedastic_rows(pkg, L, M, H):
    """full_cov = true WITH diag_var = true (src/vbmf_sparse.jl:180-182, 192-193): per-column blocks
    K_m = B' diag(sigmaVec) B + L mean(sigmaVec) SigmaB + diag(CA[m,:]) -- the weighted Gram is an extra reduction over the rows
    -- and vec(A')_m = inv(K_m) (B' diag(sigmaVec) Y)_m without a sigmaHat factor.  PARITY UNPINNED (no recorded run of this
    branch): against the oracle's dense kron(...) restatement; then the loop with both switches on."""
    Y, po = _mk_hetero(L, M, H, 90 + H, H1=2, labels=[3, 20, 25])
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    O.vbmf_sparse_(Yf, po, 3, eps=0.0, full_cov=False, diag_var=True)       # non-trivial SigmaB / CA / sigmaVecHat
    pg = _to_pkg_hetero(pkg, po)
    pkg.sparse_updateA_(Yf, pg, full_cov=True, diag_var=True); O.sparse_updateA(Yf, po, full_cov=True, diag_var=True)
    _cmp(f"full_cov+diag_var {L}x{M} H{H} updateA", pg, po, 5e-5, ("ATVecHat", "diagSigmaATVec", "SigmaA"))
    assert np.any(po.SigmaA != np.diag(np.diag(po.SigmaA)))
    assert np.all(pg.AHat[[3, 20, 25], H - 2:] == 0.0)
    pg = _to_pkg_hetero(pkg, po)
    pkg.sparse_updateB_(Yf, pg, diag_var=True); O.sparse_updateB(Yf, po, diag_var=True)     # consumes the full SigmaA
    _cmp(f"full_cov+diag_var {L}x{M} H{H} updateB", pg, po, 5e-5, ("BHat", "SigmaB"))
    if M * H <= 4000 and L < 10000:      # (at 60 000 x 26 the model prunes every column within five sweeps -- A, B -> 0 in the oracle too:
        #                                     a relative comparison of the factors is meaningless there; the single updates above are the test)
        pg = _to_pkg_hetero(pkg, po)
        d_gpu = pkg.vbmf_sparse_(Yf, pg, 5, eps=0.0, full_cov=True, diag_var=True)
        d_ref, _ = O.vbmf_sparse_(Yf, po, 5, eps=0.0, full_cov=True, diag_var=True)
        _cmp_h(f"full_cov+diag_var run5 {L}x{M} H{H}", pg, po, 2e-3)
        assert abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6


@pytest.mark.parametrize("L,M,H", [(300, 170, 5), (500, 120, 40), (400, 90, 64), (400, 84, 100), (500, 84, 160)])
def test_full_cov_against_the_oracle(pkg, L, M, H):
    """Larger shapes (all four register tilings of the per-column inverse: two columns per round up to H = 64, one for
    64 < H <= 128; the blocked Schur inverse through a global workspace for H = 160), label mask included, against the oracle's
    dense kron(...) restatement; then 6 sweeps of the loop."""
    Y, po = _mk(L, M, H, 60 + H, H1=2, labels=[3, 50, 80])
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    O.vbmf_sparse_(Yf, po, 2, eps=0.0, full_cov=False)           # a state with non-trivial SigmaB / CA
    pg = _to_pkg(pkg, po)
    pkg.sparse_updateA_(Yf, pg, full_cov=True); O.sparse_updateA(Yf, po, full_cov=True)
    _cmp(f"full_cov {L}x{M} H{H} updateA", pg, po, 5e-5, ("ATVecHat", "diagSigmaATVec", "SigmaA"))
    assert np.any(po.SigmaA != np.diag(np.diag(po.SigmaA)))     # a full matrix now
    assert np.all(pg.AHat[[3, 50, 80], H - 2:] == 0.0)
    pg = _to_pkg(pkg, po)
    pkg.sparse_updateB_(Yf, pg); O.sparse_updateB(Yf, po)        # consumes the full SigmaA
    _cmp(f"full_cov {L}x{M} H{H} updateB", pg, po, 5e-5, ("BHat", "SigmaB"))
    if M * H <= 4000:                                            # the oracle's dense inverse is O((MH)^3)
        pg = _to_pkg(pkg, po)
        pkg.vbmf_sparse_(Yf, pg, 6, eps=0.0, full_cov=True)
        O.vbmf_sparse_(Yf, po, 6, eps=0.0, full_cov=True)
        _cmp(f"full_cov {L}x{M} H{H} run6", pg, po, 1e-3)
        lb_gpu, lb_ref = pkg.lowerBound(Yf, _to_pkg(pkg, po)), O.lowerBound(Yf, po)   # the bound of a full-SigmaA state
        report(f"full_cov {L}x{M} H{H} lowerBound of the oracle's state: gpu {lb_gpu:.6f} oracle {lb_ref:.6f}")
        assert abs(lb_gpu - lb_ref) <= 2e-5 * abs(lb_ref)
    with pytest.raises(NotImplementedError):                     # H > 256: beyond every kernel of the library
        pkg._check_full_cov(True, False, 257)


@pytest.mark.parametrize("diag_var", [False, True])
def test_full_cov_wide_rank_run(pkg, diag_var):
    """The run loop with full_cov = true at 128 < H <= 256 (one 1024-thread workgroup per column through the blocked Schur inverse
    in a global workspace), either noise model: 4 sweeps at H = 150, M = 24 against the oracle's dense 3600 x 3600 inverse."""
    L, M, H = 400, 24, 150
    if diag_var:
        Y, po = _mk_hetero(L, M, H, 131, H1=2, labels=[3, 11, 20])
    else:
        Y, po = _mk(L, M, H, 131, H1=2, labels=[3, 11, 20])
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    pg = _to_pkg_hetero(pkg, po) if diag_var else _to_pkg(pkg, po)
    d_gpu = pkg.vbmf_sparse_(Yf, pg, 4, eps=0.0, full_cov=True, diag_var=diag_var)
    d_ref, _ = O.vbmf_sparse_(Yf, po, 4, eps=0.0, full_cov=True, diag_var=diag_var)
    (_cmp_h if diag_var else _cmp)(f"full_cov run4 {L}x{M} H{H} diag_var={diag_var}", pg, po, 2e-3)
    assert np.any(pg.SigmaA != np.diag(np.diag(pg.SigmaA))) and abs(d_gpu - d_ref) <= 2e-2 * d_ref + 2e-6
    assert np.all(pg.AHat[[3, 11, 20], H - 2:] == 0.0)


def test_lower_bound_trimmed(pkg):
    """lowerBoundTrimmed (src/vbmf_sparse.jl:478-489; examples/mil_util.jl:505): the device masks its M*H-long sums with
    |ATVecHat| > trim; the oracle trims the vectors like the reference and calls lowerBound.  PARITY UNPINNED (no recorded value)."""
    L, M, H = 400, 260, 6
    Y, po = _mk(L, M, H, 31)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    O.vbmf_sparse_(Yf, po, 10, eps=0.0, full_cov=False, est_cb=True)
    po.ATVecHat = po.ATVecHat.astype(np.float32).astype(np.float64)      # the device compares its fp32 copy with trim
    po.AHat = po.ATVecHat.reshape(M, H).copy()
    full = O.lowerBound(Yf, po)
    seen = set()
    for trim in (0.0, 1e-1, 0.5, 1e9):
        want = O.lowerBoundTrimmed(Yf, po, trim)
        got = pkg.lowerBoundTrimmed(Yf, _to_pkg(pkg, po), trim)
        kept = int(np.sum(np.abs(po.ATVecHat) > trim))
        report(f"sparse lowerBoundTrimmed trim={trim:g}: kept {kept} of {M * H}; gpu {got:.6f} oracle {want:.6f} (untrimmed {full:.6f})")
        assert abs(got - want) <= 1e-5 * abs(want) + 1e-3, (trim, got, want)
        seen.add(kept)
    assert len(seen) >= 3 and 0 in seen                     # the masks really differed, down to the empty one
    with pytest.raises(pkg.VbmfError):
        pkg.lowerBoundTrimmed(Yf, _to_pkg(pkg, po), -1.0)


def test_first_update_on_a_fresh_init_is_updateB(pkg):
    """A fresh vbmf_sparse_init state holds SigmaA = zeros beside diagSigmaATVec = ones (src/vbmf_sparse.jl:120-123); calling
    updateB! / updateSigma! / lowerBound BEFORE any updateA! must use that zero SigmaA, not the column sums of diagSigmaATVec."""
    L, M, H = 300, 170, 5
    Y, po = _mk(L, M, H, 52)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    po.trYTY = float(np.sum(Yf * Yf))
    assert not po.SigmaA.any() and np.all(po.diagSigmaATVec == 1.0)
    pg = _to_pkg(pkg, po)
    pkg.sparse_updateB_(Yf, pg); O.sparse_updateB(Yf, po)
    _cmp("fresh-init updateB", pg, po, 5e-5, ("BHat", "SigmaB"))
    pg = _to_pkg(pkg, po)
    pkg.sparse_updateSigma_(Yf, pg); O.sparse_updateSigma(Yf, po)
    _cmp("fresh-init updateSigma", pg, po, 5e-4, ())
    lb, lbo = pkg.lowerBound(Yf, _to_pkg(pkg, po)), O.lowerBound(Yf, po)
    assert abs(lb - lbo) <= 1e-5 * abs(lbo) + 1e-3, (lb, lbo)


def test_updates_without_Y_and_derived_constants(pkg):
    """updateCA! / updateCB! take no Y in the reference (src/vbmf_sparse.jl:284-300): the same calls work here, with or without
    a cached device copy of a matrix; a struct whose derived constants (alpha, gamma, eta) disagree with its hyper-priors is refused."""
    L, M, H = 200, 120, 4
    Y, po = _mk(L, M, H, 63)
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    Yf = Y.astype(np.float32).astype(np.float64)
    O.sparse_updateA(Yf, po, full_cov=False); O.sparse_updateB(Yf, po)
    pkg.invalidate()                                        # no cached context: the Y-free path creates one without a matrix
    pa, pb = _to_pkg(pkg, po), _to_pkg(pkg, po)
    pkg.sparse_updateCA_(pa); pkg.sparse_updateCB_(pa)
    pkg.sparse_updateCA_(pb, Y=Yf); pkg.sparse_updateCB_(pb, Y=Yf)
    O.sparse_updateCA(po); O.sparse_updateCB(po)
    for f in ("CA", "beta", "CB", "delta"):
        assert np.array_equal(getattr(pa, f), getattr(pb, f)), f
    _cmp("Y-free updateC", pa, po, 5e-5, ("CA", "beta", "CB", "delta"))
    bad = _to_pkg(pkg, po)
    bad.gamma = bad.gamma + 1.0
    with pytest.raises(ValueError, match="derived"):
        pkg.sparse_updateCB_(bad)
