"""Shared helpers for the GPU parity tests (oracle = checker only)."""
import os

import numpy as np

from oracle import vbmf_oracle as O

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")


def relF(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def to_pkg_params(pkg, po):
    """oracle params (0-based labels) -> package params (1-based labels), deep copy."""
    p = pkg.vbmf_parameters()
    p.L, p.M, p.H, p.H1 = po.L, po.M, po.H, po.H1
    p.labels = np.asarray(po.labels, dtype=np.int64) + 1
    for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB"):
        setattr(p, f, getattr(po, f).copy())
    p.sigma2 = po.sigma2
    return p


def clone_oracle(po):
    q = O.copy_params(po)
    for f in ("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB"):
        setattr(q, f, getattr(po, f).copy())
    return q


def compare(tag, pg, po, tol, fields=("AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB")):
    errs = {f: relF(getattr(pg, f), getattr(po, f)) for f in fields}
    errs["sigma2"] = abs(pg.sigma2 - po.sigma2) / abs(po.sigma2)
    report(f"{tag}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    bad = {k: v for k, v in errs.items() if not v <= tol.get(k, tol["default"])}
    assert not bad, (tag, bad, errs)
    return errs


def bf16_round(Y):
    """fp64 -> bf16 with ONE round-to-nearest-even (what the device stores), returned as float64."""
    Y = np.asarray(Y, dtype=np.float64)
    u = Y.astype(np.float32).view(np.uint32).astype(np.int64)
    base = u & 0xFFFF0000
    best = None
    for off in (-0x10000, 0, 0x10000):
        cu = base + off
        cand = (cu & 0xFFFFFFFF).astype(np.uint32).view(np.float32).astype(np.float64)
        err = np.abs(cand - Y)
        even = ((cu >> 16) & 1) == 0
        if best is None:
            best, berr, beven = cand, err, even
        else:
            take = (err < berr) | ((err == berr) & even & ~beven)
            best = np.where(take, cand, best); berr = np.where(take, err, berr); beven = np.where(take, even, beven)
    return best
