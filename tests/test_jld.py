"""The reference's on-disk log format (src/data_manip.jl:53-93: JLD = HDF5 + Julia conventions) -- CPU only.

tests/golden/ref_jld/ holds the reference's OWN recorded data files (examples/data/{vbmf_test,sparse_test}/*.jld, data
written by Julia 0.5.2 / JLD: fixtures, not source).  Checked here:
  * load_log opens those directories directly and yields exactly the arrays of the h5dump-extracted golden .npz fixtures
    (HDF5 dims reversed, time on the trailing Julia axis);
  * save_log of the same log produces files whose `h5dump` text equals the reference files' object by object (datasets,
    types, dataspaces, attributes, /_creator, the JLD encoding of `priors = Dict()`), only object addresses differ;
  * extract_params_ on the loaded reference log reproduces SURVEY App. B's spot values.
Julia itself cannot run here, so compatibility with the reference's `load` is shown structurally."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import __graft_entry__ as G

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "ref_jld")
H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"


@pytest.fixture(scope="module")
def pkg():
    p = G.load_package()
    try:
        p.data_manip.jld.lib()
    except ImportError as e:                                   # the image ships /opt/conda/lib/libhdf5.so.103
        pytest.skip(str(e))
    return p


def _dump(path):
    out = subprocess.run([H5DUMP, "-B", path], check=True, capture_output=True, text=True).stdout
    out = re.sub(r'^HDF5 ".*" \{', "HDF5 {", out, count=1)
    return re.sub(r"DATASET \d+ /_refs/", "DATASET @ /_refs/", out)      # object addresses of the two references


@pytest.mark.parametrize("name", ["vbmf_test", "sparse_test"])
def test_load_log_opens_the_reference_files(pkg, golden_dir, name):
    log, Y, priors = pkg.load_log(os.path.join(REF, name))
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    assert priors == {} and np.array_equal(Y, z["Y"])
    checked = 0
    for k in z.files:
        if k == "Y" or k == "cov_slices":
            continue
        want = np.moveaxis(z[k], 0, -1)                         # fixture is time-first (HDF5 order); Julia sees time last
        got = log[k]
        if want.shape != got.shape:                             # the fixture kept four slices of the two 40 x 40 covariances
            assert k in ("SigmaATVec", "invSigmaATVec"), (k, want.shape, got.shape)
            got = got[..., z["cov_slices"]]
        assert np.array_equal(got, want), k
        checked += 1
    assert checked >= 12
    assert log["labels"].shape == (0, 101) and log["labels"].dtype == np.int64


def test_extract_params_from_the_reference_file(pkg):
    log, _, _ = pkg.load_log(os.path.join(REF, "vbmf_test"))
    q = pkg.vbmf_parameters()
    pkg.extract_params_(log, 0, q)
    assert (q.L, q.M, q.H, q.H1) == (10, 20, 2, 0) and q.AHat.shape == (20, 2)
    assert q.AHat[0, 0] == 1.3988750154344594                   # SURVEY App. B, slice 0
    pkg.extract_params_(log, 100, q)
    assert q.sigma2 == 0.0023457154169626905 and q.BHat[0, 1] == 3.193748596202819


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="h5dump not in this image")
@pytest.mark.parametrize("name", ["vbmf_test", "sparse_test"])
def test_save_log_writes_the_reference_layout(pkg, tmp_path, name):
    log, Y, priors = pkg.load_log(os.path.join(REF, name))
    d = pkg.save_log(log, Y, priors, str(tmp_path), desc=name)
    assert sorted(os.listdir(d)) == ["inputs.jld", "log.jld"]
    with open(os.path.join(d, "log.jld"), "rb") as f:
        assert f.read(37) == b"Julia data file (HDF5), version 0.1.1"
    for fn in ("log.jld", "inputs.jld"):
        assert _dump(os.path.join(d, fn)) == _dump(os.path.join(REF, name, fn)), fn
    assert pkg.data_manip.jld.creator(os.path.join(d, "log.jld")) == pkg.data_manip.jld.creator(os.path.join(REF, name, "log.jld"))


def test_round_trip_of_a_fresh_log(pkg, tmp_path):
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((6, 4))
    p = pkg.vbmf_init(Y, 2, H1=1, labels=[1, 3], rng=rng)
    log = pkg.create_log(p)
    for t in range(2):
        p.AHat = p.AHat * 0.5; p.sigma2 = 0.3 + t
        pkg.update_log_(log, p)
    d = pkg.save_log(log, Y, {}, str(tmp_path), desc="fresh")
    log2, Y2, pri = pkg.load_log(d)
    assert pri == {} and np.array_equal(Y2, Y) and set(log2) == set(log)
    for k in log:
        assert np.array_equal(log2[k], log[k]) and log2[k].dtype == (np.int64 if log[k].dtype.kind in "iu" else np.float64), k
