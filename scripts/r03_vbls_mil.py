#!/usr/bin/env python3
"""Is fixed-basis inference (vbls!, examples/mil_util.jl:179-203) fast at the sizes its real caller uses?
The MIL classifier calls vbls!(Y, copy_vbmf_params(Y, res), 150) twice per BAG (examples/mil_util.jl:473-479; the 'dual' variant
20 iterations with full_cov = true, :518-521): Y is features x instances-of-one-bag -- L a few tens to hundreds, M a handful to a
few tens, H <= 10.  Per call and per bag, end to end (a new Y every call: session creation, upload, the loop, read-back), the
device path against the fp64 oracle on the host.   python scripts/r03_vbls_mil.py  (GPU box, repo root)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G          # noqa: E402
from oracle import vbmf_oracle as O  # noqa: E402  (the checker, timed here as the CPU side)

pkg = G.load_package()
pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)


def bag(L, M, H, rng, Bs):
    As = np.zeros((M, H)); As[np.arange(M), rng.integers(0, H, M)] = 1.0
    return Bs @ As.T + 0.05 * rng.standard_normal((L, M))


def main():
    rows = []
    for (L, Mtrain, M, H, nb) in [(166, 400, 6, 2, 40), (166, 400, 30, 5, 40), (230, 600, 60, 10, 30), (1000, 2000, 200, 10, 10)]:
        rng = np.random.default_rng(L + M)
        Bs = rng.standard_normal((L, H)) * np.linspace(1.0, 2.5, H)
        Ytr = bag(L, Mtrain, H, rng, Bs)
        res = O.vbmf_init(Ytr, H, ca=0.1, cb=0.1, sigma2=0.1, rng=np.random.default_rng(3), materialize_yhat=False)
        O.vbmf_(Ytr, res, 30, eps=0.0, est_covs=True, est_var=True)
        bags = [bag(L, M, H, rng, Bs) for _ in range(nb)]
        # --- basic model, 150 iterations (:473-479) ---
        t0 = time.perf_counter()
        for Y in bags:
            p = O.copy_vbmf_params(Y, res, rng=np.random.default_rng(1))
            O.vbls_(Y, p, 150)
        t_cpu = (time.perf_counter() - t0) / nb
        resg = pkg.vbmf_parameters()
        for f in ("L", "M", "H", "H1", "sigma2"):
            setattr(resg, f, getattr(res, f))
        resg.labels = np.zeros(0, dtype=np.int64)
        for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB"):
            setattr(resg, f, getattr(res, f).copy())
        pkg.vbls_(bags[0], pkg.copy_vbmf_params(bags[0], resg, rng=np.random.default_rng(1)), 150)      # warm the library
        # the caller's pattern: one bag at a time, the previous one garbage by then (a session whose matrix died is re-used for the
        # next matrix of its shape: __init__.py, _session_for)
        t0 = time.perf_counter()
        worst = 0.0
        for i in range(nb):
            Y = bags[i].copy()
            pg = pkg.copy_vbmf_params(Y, resg, rng=np.random.default_rng(1))
            pkg.vbls_(Y, pg, 150)
            del Y
        t_gpu = (time.perf_counter() - t0) / nb
        po = O.copy_vbmf_params(bags[-1], res, rng=np.random.default_rng(1)); O.vbls_(bags[-1], po, 150)
        worst = float(np.linalg.norm(pg.AHat - po.AHat) / np.linalg.norm(po.AHat))
        rows.append((f"vbls! x150, basic   L={L} M={M} H={H}", t_cpu, t_gpu, worst))
        # --- ARD-sparse model, 20 iterations with full_cov = true (the 'dual' classifier, :518-521) ---
        so = O.vbmf_sparse_init(Ytr, H, ca=1.0, cb=1.0, sigma=1.0, rng=np.random.default_rng(7), full_cov=False, materialize_yhat=False)
        O.vbmf_sparse_(Ytr, so, 12, eps=0.0, full_cov=False)
        nbf = max(2, nb // 4) if M * H > 1000 else nb            # the reference's own MH x MH inverse: seconds per call at M*H = 2000

        def cpu_full(Y):
            q = O.copy_vbmf_params(Y, so, rng=np.random.default_rng(1))
            for _ in range(20):                                  # examples/mil_util.jl:186-189 with full_cov = true
                O.sparse_updateA(Y, q, full_cov=True)
                O.sparse_updateCA(q)
                O.sparse_updateSigma(Y, q)
            return q
        t0 = time.perf_counter()
        for Y in bags[:nbf]:
            qo = cpu_full(Y)
        t_cpu = (time.perf_counter() - t0) / nbf
        sg = pkg.vbmf_sparse_parameters()
        for f in ("L", "M", "H", "MH", "H1", "alpha0", "beta0", "alpha", "gamma0", "delta0", "gamma", "sigmaHat", "eta0", "zeta0",
                  "eta", "zeta", "trYTY"):
            setattr(sg, f, getattr(so, f))
        sg.labels = np.asarray(so.labels, dtype=np.int64) + 1
        for f in ("AHat", "ATVecHat", "diagSigmaATVec", "SigmaA", "BHat", "SigmaB", "CA", "beta", "CB", "delta"):
            setattr(sg, f, getattr(so, f).copy())
        pkg.vbls_(bags[0], pkg.copy_vbmf_params(bags[0], sg, rng=np.random.default_rng(1)), 20, full_cov=True)
        t0 = time.perf_counter()
        for i in range(nb):
            Y = bags[i].copy()
            qg = pkg.copy_vbmf_params(Y, sg, rng=np.random.default_rng(1))
            pkg.vbls_(Y, qg, 20, full_cov=True)
            del Y
        t_gpu = (time.perf_counter() - t0) / nb
        qo = cpu_full(bags[-1])
        worst = float(np.linalg.norm(qg.AHat - qo.AHat) / np.linalg.norm(qo.AHat))
        rows.append((f"vbls! x20, sparse full_cov L={L} M={M} H={H}", t_cpu, t_gpu, worst))
    print(f"{'call':46s} {'oracle (host) ms':>18s} {'device ms':>12s} {'ratio':>8s} {'rel.err AHat':>14s}")
    for name, tc, tg, w in rows:
        print(f"{name:46s} {tc * 1e3:18.3f} {tg * 1e3:12.3f} {tc / tg:8.2f} {w:14.2e}")


if __name__ == "__main__":
    main()
