"""Worker of tests/test_gpu_two_ranks.py: one rank of a row-sharded run (one process per rank).

    python tests/two_rank_worker.py RANK WORLD PORT OUTDIR TRANSPORT CASE

TRANSPORT = "host": the library's all-reduces go through vbmf_comm_set_transport + a gloo all-reduce on host
copies, so that several ranks can share ONE GPU; "rccl": the real communicator, one GPU per rank."""
import os
import sys

import numpy as np


def problem(L, M, H, seed):
    """Common to all ranks and to the single-context reference run in the parent (seeded)."""
    rng = np.random.default_rng(seed)
    Hs = H                                  # latent rank = model rank, well separated scales: a well-conditioned fixed point
    Bs = rng.standard_normal((L, Hs)) * np.linspace(1.0, 3.0, Hs)
    As = np.zeros((M, Hs))
    As[np.arange(M), rng.integers(0, Hs, M)] = 1.0
    Y = Bs @ As.T + 0.05 * rng.standard_normal((L, M))
    A0, B0 = rng.standard_normal((M, H)), rng.standard_normal((L, H))
    return Y, A0, B0


# name -> (L, M, H): L deliberately not a multiple of the rank count.  "h128" runs the H = 128 kernels of the
# 8-GPU BASELINE configuration (16 accumulator tiles per wave, un-fused post/Gram), "h200" the H > 128 control path.
# "sparse" / "hetero": the ARD-sparse variant (homoscedastic / one noise precision per row) row-sharded.
# "trial": the three-group variant with its hyper-prior fits (replicated M x H work: no further collective).
# "straddle": the two shards (2049 / 2048 rows) sit on either side of the narrow-geometry threshold of the H <= 32 kernel;
# the ranks must still agree on the padded width of the all-reduced Y'B partial (the decision uses the nominal shard size).
CASES = {"h12": (1531, 700, 12), "h128": (1203, 520, 128), "h200": (901, 420, 200), "sparse": (1101, 480, 6),
         "hetero": (1101, 480, 6), "hetero_stop": (1101, 480, 6), "trial": (1101, 480, 6), "straddle": (4097, 2100, 12)}
# "hetero_stop": the heteroscedastic loop ends EARLY on the device-side stop flag (eps > 0) with sweeps still enqueued behind
# it, then runs one more updateA! / updateB! on the state the loop left: every collective enqueued after the stop must leave
# the frozen state -- the all-reduced Grams included -- exactly as the last executed sweep produced it.
EPS_CASE = {"hetero_stop": float(os.environ.get("VBMF_TEST_EPS", "0.02"))}     # the parent picks eps in a wide gap of the d sequence
SPARSE_CASES = ("sparse", "hetero", "hetero_stop", "trial")
TRIAL_H0, TRIAL_M0 = 4, 190
EPS, SEED = 0.0, 4242
NITERS = {"h12": 12, "h128": 5, "h200": 5, "sparse": 8, "hetero": 8, "hetero_stop": 14, "trial": 8, "straddle": 4}
HYPER = dict(alpha0=1e-10, beta0=1e-10, gamma0=1e-10, delta0=1e-10, eta0=1e-10, zeta0=1e-10)


def variant_of(pkg, case):
    return {"sparse": pkg.capi.VBMF_VARIANT_SPARSE_DIAG, "hetero": pkg.capi.VBMF_VARIANT_SPARSE_DIAGVAR,
            "hetero_stop": pkg.capi.VBMF_VARIANT_SPARSE_DIAGVAR,
            "trial": pkg.capi.VBMF_VARIANT_TRIAL_DIAG}.get(case, pkg.capi.VBMF_VARIANT_BASIC)


def run_sparse(pkg, ctx, Y, A0, B0, H, niter, hetero, L_global, row0, trial=False, eps=EPS, after_stop=False):
    """vbmf_sparse_init's initial state (src/vbmf_sparse.jl:101-153, ca = cb = sigma = 1) on this rank's rows."""
    M = A0.shape[0]
    n = Y.shape[0]
    ctx.set_Y(Y)
    ctx.sparse_set_state(A0.reshape(M * H), np.ones(M * H), np.ones(M * H), 1e-10 * np.ones(M * H), B0, np.zeros((H, H)),
                         np.ones(H), 1e-10 * np.ones(H), 1.0, 1e-10, HYPER)
    if hetero:
        ctx.sparse_set_noise_rows(np.ones(n), 1e-10 * np.ones(n), 1e-10 + M / 2)
    if trial:
        pri = {k: (1e-10 + 0.5 if k in ("alpha1", "alpha2", "alpha3") else 1e-10) for k in pkg.capi.Context.TRIAL_KEYS}
        ctx.trial_set_priors(TRIAL_H0, TRIAL_M0, pri)
        it, d, _ = ctx.trial_run(niter, eps=EPS, est_cb=True, est_priors=True)
    else:
        it, d, _ = ctx.sparse_run(niter, eps=eps, est_cb=True)
    if after_stop:                                   # one more A and B update on the state the (early-stopped) loop left
        ctx.sparse_step(pkg.capi.SSTEP_A)
        ctx.sparse_step(pkg.capi.SSTEP_B)
    s = ctx.sparse_get_state()
    out = dict(it=it, d=d, trYY=ctx.trYY(), AHat=s["ATVecHat"].reshape(M, H), BHat=s["BHat"], SigmaA=np.diag(s["SigmaA_diag"]),
               SigmaB=s["SigmaB"], CA_diag=s["CA"], CB_diag=s["CB"], sigma2=s["sigmaHat"], trace=np.zeros(1),
               elbo=0.0 if hetero else ctx.sparse_lower_bound())
    if hetero:
        out["sigmaVecHat"], out["zetaVec"] = ctx.sparse_get_noise_rows()
    if trial:
        out["priors"] = np.array(list(ctx.trial_get_priors()[2].values()))
    return out



def run(pkg, ctx, Y, A0, B0, H, niter):
    z = np.zeros((H, H))
    ctx.set_Y(Y)
    ctx.set_state(A0, B0, z, z, 0.1 * np.ones(H), 0.1 * np.ones(H), 0.1)
    it, d, tr = ctx.run(niter, eps=EPS, est_covs=True, est_var=True, want_trace=True)
    s = ctx.get_state()
    return dict(it=it, d=d, trace=tr[:, :3], trYY=ctx.trYY(), elbo=ctx.elbo(), **s)


def main():
    rank, world, port, outdir, transport = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    case = sys.argv[6]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as G
    pkg = G.load_package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    L, M, H = CASES[case]
    Y, A0, B0 = problem(L, M, H, SEED)
    r0, n = pkg.dist.row_shard(L, world, rank)
    dev = rank if transport == "rccl" else 0
    with pkg.capi.Context(n, M, H, y_dtype=pkg.VBMF_Y_BF16, device=dev, nranks=world, rank=rank, L_global=L,
                          row_offset=r0, variant=variant_of(pkg, case)) as ctx:
        if transport == "rccl":
            uid = [pkg.capi.Context.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(uid[0])
        else:
            ctx.comm_set_transport(pkg.dist.host_staged_transport(lambda a: dist.all_reduce(torch.from_numpy(a))))
        if case in SPARSE_CASES:
            res = run_sparse(pkg, ctx, Y[r0:r0 + n], A0, B0[r0:r0 + n], H, NITERS[case], case.startswith("hetero"), L, r0, case == "trial",
                             eps=EPS_CASE.get(case, EPS), after_stop=case in EPS_CASE)
        else:
            res = run(pkg, ctx, Y[r0:r0 + n], A0, B0[r0:r0 + n], H, NITERS[case])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), row0=r0, nrows=n, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
