"""Trajectory log (src/data_manip.jl:6-118) -- host-side twin in vbmatrixfactorization.jl_amd/data_manip.py.
CPU-only: structure of the log, round trip through disk (the on-disk container is the reference's JLD/HDF5 layout,
see tests/test_jld.py for the container itself), and that the arrays of the reference's OWN recorded log
(examples/data/vbmf_test/log.jld, here as the npz extract tests/golden/vbmf_test.npz) are readable with extract_params_."""
import numpy as np
import pytest

import __graft_entry__ as G


@pytest.fixture(scope="module")
def pkg():
    return G.load_package()


def test_log_structure_and_round_trip(pkg, tmp_path):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((7, 5))
    p = pkg.vbmf_init(Y, 3, ca=0.1, cb=0.1, sigma2=0.1, H1=1, labels=[2, 4], rng=rng)
    log = pkg.create_log(p)
    assert log["AHat"].shape == (5, 3) and log["sigma2"].shape == (1,) and log["L"].tolist() == [7]
    states = [p.AHat.copy()]
    for t in range(3):
        p.AHat = p.AHat + 1.0; p.sigma2 = 0.1 * (t + 2); p.BHat = p.BHat * 2.0
        states.append(p.AHat.copy())
        pkg.update_log_(log, p)
    # time is the trailing axis (src/data_manip.jl:41); scalars are vectors
    assert log["AHat"].shape == (5, 3, 4) and log["BHat"].shape == (7, 3, 4) and log["YHat"].shape == (7, 5, 4)
    assert log["sigma2"].shape == (4,) and log["labels"].shape == (2, 4) and log["H1"].tolist() == [1, 1, 1, 1]
    d = pkg.save_log(log, Y, {}, str(tmp_path), desc="run1")
    log2, Y2, priors = pkg.load_log(d)
    assert np.array_equal(Y2, Y) and priors == {} and set(log2) == set(log)
    q = pkg.vbmf_parameters()
    for t in range(4):
        pkg.extract_params_(log2, t, q)
        assert np.array_equal(q.AHat, states[t]) and q.L == 7 and q.H1 == 1 and isinstance(q.sigma2, float)
        assert np.array_equal(q.labels, p.labels)
    assert q.sigma2 == pytest.approx(0.4)
    with pytest.raises(RuntimeError, match="does not contain any log files"):
        pkg.load_log(str(tmp_path))
    # default description = date_time (src/data_manip.jl:55-57)
    d2 = pkg.save_log(log, Y, {}, str(tmp_path))
    assert d2 != d and len(d2.rsplit("/", 1)[1]) == 15


def test_reference_recorded_log_is_readable(pkg, golden_dir):
    """The reference's recorded log (HDF5 dims are Julia's reversed, so the fixture is time-first): put time last, as
    Julia sees it, and read slices with extract_params_."""
    z = np.load(f"{golden_dir}/vbmf_test.npz")
    log = {k: np.moveaxis(z[k], 0, -1) for k in z.files if k != "Y"}
    q = pkg.vbmf_parameters()
    pkg.extract_params_(log, 0, q)
    assert (q.L, q.M, q.H, q.H1) == (10, 20, 2, 0)
    assert q.AHat.shape == (20, 2) and q.AHat[0, 0] == 1.3988750154344594          # SURVEY App. B, slice 0
    pkg.extract_params_(log, 100, q)
    assert q.sigma2 == 0.0023457154169626905 and q.BHat[0, 1] == 3.193748596202819  # SURVEY App. B, slice 100
