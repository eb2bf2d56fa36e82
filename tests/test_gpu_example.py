"""BASELINE config 1 (200 x 100, rank 5, the examples/toy_data.jl plumbing): the example script's flow on the GPU --
vbmf_init, vbmf with logging, load_log / extract_params_, vbmf_sparse -- against the oracle from the same initial state."""
import copy
import os
import sys

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import vbmf_oracle as O
from tests.helpers import relF, report

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_toy_data_example_config1(tmp_path):
    G.build()
    pkg = G.load_package()
    pkg.set_defaults(y_dtype=pkg.VBMF_Y_F32, factor_dtype=pkg.VBMF_FACTOR_AUTO)
    import toy_data
    L, M, H = 200, 100, 5
    out = toy_data.main(L, M, H, seed=7, data_path=str(tmp_path), quiet=True)
    # the factorization explains the data down to the noise (0.05 * sqrt(L) ~ spectral norm of the noise matrix's scale)
    assert out["err"] < 0.05 * (np.sqrt(L) + np.sqrt(M)) * 1.5
    # the same run in the oracle, from the same initial parameters (vbmf leaves params_init untouched)
    p0 = out["params_init"]
    po = O.vbmf_parameters()
    po.L, po.M, po.H, po.H1 = L, M, H, 0
    po.labels = np.zeros(0, dtype=np.int64)
    for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB"):
        setattr(po, f, np.array(getattr(p0, f), copy=True))
    po.sigma2 = p0.sigma2
    Yf = out["Y"].astype(np.float32).astype(np.float64)
    O.vbmf_(Yf, po, 100, eps=1e-6, est_covs=True, est_var=True)
    res = out["res"]
    errs = dict(A=relF(res.AHat, po.AHat), B=relF(res.BHat, po.BHat), s2=abs(res.sigma2 - po.sigma2) / po.sigma2)
    report(f"config 1 (toy_data example, 200x100 H=5, 100 sweeps, eps=1e-6): " + " ".join(f"{k}={v:.2e}" for k, v in errs.items())
           + f" ||Y-YHat||_2 = {out['err']:.4f}")
    assert errs["A"] < 2e-3 and errs["B"] < 2e-3 and errs["s2"] < 5e-3, errs
    # the log holds the initial state + one slice per executed sweep, and slice 3's error is larger than the final one
    T = out["log"]["sigma2"].shape[0]
    assert 2 <= T <= 101 and out["log"]["AHat"].shape == (M, H, T)
    assert out["err_it"] >= out["err"] * 0.999
    assert np.isfinite(out["err_sparse"])
