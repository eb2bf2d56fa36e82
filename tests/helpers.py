"""Shared helpers for the GPU parity tests (oracle = checker only)."""
import os
import sys

import numpy as np

from oracle import vbmf_oracle as O

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


BASELINE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpu_parity_baseline.json")
# a measured error may move by rounding luck when a kernel's summation order changes; below this it is fp32 storage noise
BASELINE_FLOOR = 3e-6
BASELINE_FACTOR = 3.0
_baseline = None


def baseline_guard(line):
    """Regression guard under every comparison: `line` is a report line `<tag>: field=err ...`; each error must stay within
    max(3 x the figure measured on an MI355X for that tag and field, BASELINE_FLOOR) (tests/golden/make_parity_baseline.py)."""
    global _baseline
    if _baseline is None:
        import json
        _baseline = json.load(open(BASELINE)) if os.path.exists(BASELINE) else {}
    sys.path.insert(0, os.path.dirname(BASELINE))
    try:
        from make_parity_baseline import parse
    finally:
        sys.path.pop(0)
    p = parse(line)
    if p is None or p[0] not in _baseline:
        return
    tag, errs = p
    bad = {k: (v, _baseline[tag][k]) for k, v in errs.items()
           if k in _baseline[tag] and not v <= max(BASELINE_FACTOR * _baseline[tag][k], BASELINE_FLOOR)}
    assert not bad, (f"regression against the measured baseline of '{tag}' (field: (now, measured)); if the change is intended, "
                     f"regenerate tests/golden/gpu_parity_baseline.json", bad)


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")
    baseline_guard(line)


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
