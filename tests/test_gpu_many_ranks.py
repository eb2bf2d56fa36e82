"""EIGHT row shards of one problem on ONE GPU, H = 128: the kernels and the collective code path of BASELINE.json's 8-GPU
configuration (config 4: 16 accumulator tiles per wave, un-fused post / Gram kernels, fragment-major products), with the
replicated state compared across all eight ranks, against a single-rank run of the same library AND against the fp64
oracle.

The GPU box admits at most 6 processes on its card, so eight ranks cannot be eight processes there; each rank is a THREAD
of this process with its own context (distinct contexts are independent, include/vbmf_hip.h), and the library's all-reduce
goes through vbmf_comm_set_transport to an in-process reduction over the threads (host-staged, summed in rank order, so all
ranks receive bit-identical sums -- what RCCL's all-reduce guarantees too).  Everything else is the production path:
row-sharded tiling, L_global in the H x H algebra, the out-of-place reductions of Y'B and of [B'B | dB'dB | tr(B'YA)],
the device-side stop flag.  L is not a multiple of the rank count."""
import threading

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


class ThreadAllReduce:
    """sum over `world` threads, fixed rank order; every rank gets the same bits"""

    def __init__(self, world):
        self.world = world
        self.slots = [None] * world
        self.bar = threading.Barrier(world)

    def make(self, rank):
        def fn(a):
            self.slots[rank] = a.copy()
            self.bar.wait(timeout=120)
            tot = self.slots[0].copy()
            for r in range(1, self.world):
                tot += self.slots[r]
            self.bar.wait(timeout=120)
            a[:] = tot
        return fn


def _problem(L, M, H, seed):
    rng = np.random.default_rng(seed)
    Bs = rng.standard_normal((L, H)) * np.linspace(1.0, 3.0, H)
    As = np.zeros((M, H))
    As[np.arange(M), rng.integers(0, H, M)] = 1.0
    Y = Bs @ As.T + 0.05 * rng.standard_normal((L, M))
    return Y, rng.standard_normal((M, H)), rng.standard_normal((L, H))


def _run(c, Y, A0, B0, H, niter):
    z = np.zeros((H, H))
    c.set_Y(Y)
    c.set_state(A0, B0, z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
    it, d, tr = c.run(niter, eps=0.0, est_covs=True, est_var=True, want_trace=True)
    s = c.get_state()
    return dict(it=it, d=d, trace=tr[:, :3].copy(), trYY=c.trYY(), elbo=c.elbo(), Ys=c.get_Y(), **s)


@pytest.mark.parametrize("world,L,M,H,niter", [(8, 4099, 1040, 128, 3), (5, 2603, 700, 40, 4)])
def test_many_ranks_one_gpu_threads(pkg, world, L, M, H, niter):
    Y, A0, B0 = _problem(L, M, H, 9090 + H)
    with pkg.capi.Context(L, M, H, y_dtype=pkg.VBMF_Y_BF16) as c:
        ref = _run(c, Y, A0, B0, H, niter)
    ar = ThreadAllReduce(world)
    out, errs_t = [None] * world, []

    def rank_main(r):
        try:
            r0, n = pkg.dist.row_shard(L, world, r)
            with pkg.capi.Context(n, M, H, y_dtype=pkg.VBMF_Y_BF16, nranks=world, rank=r, L_global=L, row_offset=r0) as c:
                c.comm_set_transport(pkg.dist.host_staged_transport(ar.make(r)))
                res = _run(c, Y[r0:r0 + n], A0, B0[r0:r0 + n], H, niter)
                res["row0"] = r0
                out[r] = res
        except Exception as e:                                 # a failed rank must not leave the others at the barrier
            errs_t.append((r, repr(e)))
            ar.bar.abort()

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errs_t, errs_t
    assert all(o is not None for o in out)
    # replicated quantities: bit-identical on all ranks
    for k in ("AHat", "SigmaA", "SigmaB", "CA_diag", "CB_diag", "sigma2", "d", "trace", "trYY", "it"):
        for o in out[1:]:
            assert np.array_equal(out[0][k], o[k]), k
    assert [o["row0"] for o in out] == [pkg.dist.row_shard(L, world, r)[0] for r in range(world)]
    B = np.concatenate([o["BHat"] for o in out], axis=0)
    Ys = np.concatenate([o["Ys"] for o in out], axis=0)
    assert np.array_equal(Ys, ref["Ys"])                       # the shards tile the same stored matrix
    assert out[0]["it"] == ref["it"] == niter

    def errs_vs(AHat, BHat, SA, SB, ca, cb, s2, d, elbo):
        return dict(A=relF(out[0]["AHat"], AHat), B=relF(B, BHat), SA=relF(out[0]["SigmaA"], SA), SB=relF(out[0]["SigmaB"], SB),
                    ca=relF(out[0]["CA_diag"], ca), cb=relF(out[0]["CB_diag"], cb), s2=abs(float(out[0]["sigma2"]) - s2) / s2,
                    d=abs(float(out[0]["d"]) - d) / d, elbo=abs(float(out[0]["elbo"]) - elbo) / abs(elbo))

    e1 = errs_vs(ref["AHat"], ref["BHat"], ref["SigmaA"], ref["SigmaB"], ref["CA_diag"], ref["CB_diag"], ref["sigma2"], ref["d"], ref["elbo"])
    report(f"{world} ranks (threads, one GPU) vs 1 rank, {L}x{M} H={H}, {niter} sweeps: " + " ".join(f"{k}={v:.2e}" for k, v in e1.items()))
    # the sharded run takes the split-pass kernels (post_gram2 on summed slabs), the single rank may take the register epilogue:
    # same arithmetic, but in the bf16 factor modes each rounds its three-term bf16 product (2^-17) differently
    assert max(e1[k] for k in ("A", "B", "ca", "cb")) < 6e-5, e1
    assert max(e1[k] for k in ("SA", "SB", "s2")) < 5e-4 and e1["elbo"] < 1e-4 and e1["d"] < 2e-2, e1

    # ... and against the fp64 oracle on the matrix as stored (not a self-comparison)
    po = O.vbmf_parameters()
    po.L, po.M, po.H, po.H1 = L, M, H, 0
    po.labels = np.zeros(0, dtype=np.int64)
    po.AHat, po.BHat = A0.copy(), B0.copy()
    po.SigmaA = np.zeros((H, H)); po.SigmaB = np.zeros((H, H))
    po.CA = 0.1 * np.eye(H); po.CB = 0.1 * np.eye(H); po.invCA = 10 * np.eye(H); po.invCB = 10 * np.eye(H)
    po.sigma2 = 0.1
    otr = []
    O.vbmf_(Ys, po, niter, eps=0.0, est_covs=True, est_var=True, fused=True, trace=otr)
    e2 = errs_vs(po.AHat, po.BHat, po.SigmaA, po.SigmaB, np.diag(po.CA), np.diag(po.CB), po.sigma2, otr[-1][0], otr[-1][2])
    report(f"{world} ranks (threads, one GPU) vs fp64 ORACLE, {L}x{M} H={H}, {niter} sweeps: " + " ".join(f"{k}={v:.2e}" for k, v in e2.items()))
    assert max(e2[k] for k in ("A", "B", "SA", "SB", "ca", "cb")) < 2e-4, e2          # bf16x2 path: 1e-4 per update
    assert e2["s2"] < 1e-3 and e2["d"] < 2e-2 and e2["elbo"] < 1e-4, e2


def test_error_on_one_rank_stops_every_rank(pkg):
    """A device error on ONE rank of a row-sharded run must stop EVERY rank, at the same sweep, with the same error class:
    the ranks' error flags travel as one more number of the packed Gram message ([B'B | dB'dB | tr(B'YA) | flag], summed by the
    same all-reduce) and every rank's loop test reads the sum.  Rank 1's Y*A pass is made to give up its in-launch hand-off
    (VBMF_DEBUG_EPI_EXPECT_SKEW with a short spin limit); before round 3 rank 0 ran on into the all-reduces with rank 1's
    garbage partials and returned OK."""
    world, n, M, H, niter = 2, 70000, 1100, 48, 6               # per-rank shard = the shape of test_epilogue_handoff_timeout_is_reported
    L = world * n
    rng = np.random.default_rng(77)
    A0, B0 = rng.standard_normal((M, H)), rng.standard_normal((L, H))
    z = np.zeros((H, H))
    ar = ThreadAllReduce(world)
    out, errs_t = [None] * world, []

    def rank_main(r):
        try:
            with pkg.capi.Context(n, M, H, y_dtype=pkg.VBMF_Y_BF16, nranks=world, rank=r, L_global=L, row_offset=r * n) as c:
                dims = c.dims()
                c.comm_set_transport(pkg.dist.host_staged_transport(ar.make(r)))
                c.set_Y_synthetic(11, H, 0.05)
                c.set_state(A0, B0[r * n:(r + 1) * n], z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
                if r == 1:
                    c.debug_set(pkg.capi.DEBUG_EPI_SPIN_LIMIT, 2000)
                    c.debug_set(pkg.capi.DEBUG_EPI_EXPECT_SKEW, 1)
                try:
                    c.run(niter, eps=0.0, est_covs=True, est_var=True)
                    out[r] = ("ok", None, dims)
                except pkg.VbmfError as e:
                    out[r] = ("err", e, dims)
        except Exception as e:                                 # a failed rank must not leave the other at the barrier
            errs_t.append((r, repr(e)))
            ar.bar.abort()

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errs_t, errs_t
    dims = out[0][2]
    if dims["nsplit2"] != 1 or dims["NH"] > 2 or dims["narrow"]:
        pytest.skip(f"planner did not choose the un-split pass (register epilogue) here: {dims}")
    assert out[0][0] == "err" and out[1][0] == "err", out       # BOTH ranks report the failure
    for r in range(world):
        assert out[r][1].code == pkg.capi.VBMF_ERR_SYNC, out[r][1]
    assert "another rank" in str(out[0][1]) and "another rank" not in str(out[1][1]), (str(out[0][1]), str(out[1][1]))
