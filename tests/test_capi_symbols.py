"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
that include/vbmf_hip.h declares; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import __graft_entry__ as G

ROOT = G.ROOT


@pytest.fixture(scope="module")
def pkg():
    G.build()
    return G.load_package()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vbmf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vbmf_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_header_symbols_exported(pkg):
    lib = ctypes.CDLL(pkg.capi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vbmf_hip.h but not exported"
    assert sorted(pkg.capi.SYMBOLS) == declared


def test_opts_struct_matches_header(pkg):
    o = pkg.capi.VbmfOpts()
    pkg.capi.lib().vbmf_default_opts(ctypes.byref(o))
    assert o.struct_size == ctypes.sizeof(pkg.capi.VbmfOpts) == 56
    assert o.nranks == 1 and o.y_dtype == pkg.capi.VBMF_Y_BF16 and o.reference_compat == 0xFFFFFFFF


def test_no_cpu_fallback(pkg):
    """Without a HIP device vbmf_create must fail with a clear message -- never compute on the CPU."""
    try:
        ctx = pkg.capi.Context(64, 32, 4)
    except pkg.VbmfError as err:
        e = err
    else:
        ctx.close()
        pytest.skip("GPU present")
    assert e.code in (-2, -3)
    assert "no CPU fallback" in str(e) or "HIP" in str(e)
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((20, 10))
    p = pkg.vbmf_init(Y, 2, rng=rng)
    with pytest.raises(pkg.VbmfError):
        pkg.vbmf_(Y, p, 3)


def test_product_does_not_import_oracle():
    """The product path must not route through oracle/ (or any CPU implementation)."""
    pkg_dir = G.PKG_DIR
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".jl", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "vbmf_oracle" not in txt, (dirpath, f)


def test_host_init_matches_reference_layout(pkg):
    """vbmf_init field semantics (src/vbmf.jl:48-73): shapes, zeroed mask block, C = c*I."""
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((12, 9))
    p = pkg.vbmf_init(Y, 4, ca=0.5, cb=2.0, sigma2=0.3, H1=2, labels=[1, 9], rng=rng)
    assert (p.L, p.M, p.H, p.H1) == (12, 9, 4, 2)
    assert p.AHat.shape == (9, 4) and p.BHat.shape == (12, 4)
    assert np.all(p.AHat[[0, 8], 2:] == 0) and np.all(p.AHat[[0, 8], :2] != 0)
    assert np.array_equal(p.CA, 0.5 * np.eye(4)) and np.allclose(p.invCB, np.eye(4) / 2.0)
    assert np.all(p.SigmaA == 0) and p.sigma2 == 0.3
    assert np.allclose(p.YHat, p.BHat @ p.AHat.T)
    q = pkg.copy(p)
    assert q.AHat is p.AHat and q.CA is p.CA          # shallow, like src/vbmf.jl:80-88
    with pytest.raises(IndexError):
        pkg.vbmf_init(Y, 4, H1=1, labels=[0], rng=rng)  # labels are 1-based
