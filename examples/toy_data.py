"""The reference's examples/toy_data.jl, line for line, on the MI355X path (Python host over the C ABI).

    python examples/toy_data.py [L M H]          # defaults 10 20 2 like the reference; 200 100 5 = BASELINE config 1

Differences from the Julia script, all stated where they occur: logs are .npz instead of JLD; the sparse run uses
full_cov=False (the branch the accelerated path builds -- and the one the reference's own study forces,
examples/mil_util.jl:122); extract_params_ takes a 0-based slice index."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402


def toy_matrix(L, M, H, std, rng):
    """examples/toy_data.jl:7-18."""
    B = rng.standard_normal((L, H))                       # B is a random matrix whose columns we select from
    A = np.zeros((M, H))                                  # A is the sparse matrix that selects the columns
    A[np.arange(M), rng.integers(0, H, M)] = 1.0
    Y = B @ A.T + std * rng.standard_normal((L, M))
    return Y, A, B


def main(L=10, M=20, H=2, seed=0, data_path=None, quiet=False):
    VBMF = G.load_package()
    say = (lambda *a: None) if quiet else print
    rng = np.random.default_rng(seed)
    Y_toy, A_toy, B_toy = toy_matrix(L, M, H, 0.05, rng)
    data_path = data_path or tempfile.mkdtemp(prefix="vbmf_data_")

    say(" ----------- Basic VB Matrix factorization ---------------- \n")
    params_init = VBMF.vbmf_init(Y_toy, H, ca=0.1, cb=0.1, sigma2=0.1, rng=rng)
    res_vbmf = VBMF.vbmf(Y_toy, params_init, 100, est_covs=True, est_var=True, verb=not quiet, logdir=data_path,
                         desc="vbmf_test")
    err = np.linalg.norm(Y_toy - res_vbmf.YHat, 2)        # Julia 0.5 norm(::Matrix) is the spectral norm
    say(f"||Y - Yhat|| = {err}")

    # this is how the log is loaded and used
    vbmf_log, data, priors = VBMF.load_log(os.path.join(data_path, "vbmf_test"))
    params_vbmf = VBMF.vbmf_parameters()                  # dummy variable to store individual step data
    it = 3
    time_slice = VBMF.extract_params_(vbmf_log, it, params_vbmf)
    # the logged YHat is the INITIAL product in every slice, in the reference too (src/vbmf.jl:217 refreshes it only
    # after the loop); the error at iteration `it` is therefore formed from the factors
    err_it = np.linalg.norm(Y_toy - time_slice.BHat @ time_slice.AHat.T, 2)
    say(f"in iteration number {it}, the error was ||Y - Yhat|| = {err_it}\n")

    say(" ----------- VB Matrix factorization with sparse A ---------------- \n")
    params_sparse_init = VBMF.vbmf_sparse_init(Y_toy, H, ca=0.1, cb=0.1, sigma=0.1, rng=rng)
    params_sparse_init.AHat = A_toy                       # (has no numerical effect: SURVEY App. A QS7)
    sparse_vbmf, d = VBMF.vbmf_sparse(Y_toy, params_sparse_init, 100, diag_var=False, verb=not quiet, logdir=data_path,
                                      desc="sparse_test", full_cov=False)
    err_sparse = np.linalg.norm(Y_toy - sparse_vbmf.BHat @ sparse_vbmf.AHat.T, 2)
    say(f"||Y - Yhat|| = {err_sparse}\n")

    say(" original A          basic reconstruction          sparse reconstruction with true start")
    for m in range(min(M, 20)):
        say(A_toy[m, :], "   ", np.round(res_vbmf.AHat[m, :], 4), "   ", np.round(sparse_vbmf.AHat[m, :], 4))
    return dict(Y=Y_toy, A=A_toy, B=B_toy, params_init=params_init, res=res_vbmf, sparse=sparse_vbmf, err=err,
                err_it=err_it, err_sparse=err_sparse, log=vbmf_log, data_path=data_path)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]]
    main(*a)
