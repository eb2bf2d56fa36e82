#!/usr/bin/env python3
"""Extract the reference's recorded trajectories into small .npz fixtures.

Source (read-only, never executed): the four HDF5/JLD data files the reference ships under
  /root/reference/examples/data/vbmf_test/{inputs,log}.jld
  /root/reference/examples/data/sparse_test/{inputs,log}.jld
They were written by the reference itself (src/data_manip.jl:6-66, driven by
examples/toy_data.jl:33-36 and :53-56) with Julia 0.5.2, and are the only pinned numbers the
reference holds for this path.

Only data is copied (arrays of float64 / int64); no reference source text is stored.
Reading uses the HDF5 command line tool `h5dump -b LE` (no h5py in this image).  HDF5 dims are the
Julia dims reversed and the raw bytes are Julia column-major, so every array is read with the HDF5
shape in C order and then transposed to (t, <julia dims...>) -- time first.

Dropped to keep the fixtures small: YHat (constant over the log because vbmf! refreshes it only
after the loop, src/vbmf.jl:217) and all but slices {0,1,2,100} of the two 40x40 sparse covariances.

Run:  python tests/golden/extract_fixtures.py      (in the build container; needs /root/reference)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference/examples/data"
H5DUMP = "/opt/conda/bin/h5dump"
H5LS = "/opt/conda/bin/h5ls"
OUT = os.path.dirname(os.path.abspath(__file__))


def _datasets(path):
    out = subprocess.run([H5LS, "-r", path], check=True, capture_output=True, text=True).stdout
    ds = {}
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 3 and parts[1] == "Dataset" and not parts[0].startswith("/_"):
            name = parts[0].lstrip("/")
            dims = line[line.index("{") + 1: line.index("}")]
            if dims in ("NULL", "SCALAR"):
                continue
            ds[name] = tuple(int(x) for x in dims.split(","))
    return ds


def _raw(path, name, dtype):
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        tmp = f.name
    try:
        subprocess.run([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", tmp, path],
                       check=True, capture_output=True)
        return np.fromfile(tmp, dtype=dtype)
    finally:
        os.unlink(tmp)


INT_FIELDS = {"L", "M", "H", "H1", "MH"}


def read_log(path):
    res = {}
    for name, h5shape in _datasets(path).items():
        dtype = "<i8" if name in INT_FIELDS else "<f8"
        a = _raw(path, name, dtype).reshape(h5shape)
        # (t, reversed julia dims) -> (t, julia dims)
        if a.ndim >= 2:
            a = a.transpose((0,) + tuple(range(a.ndim - 1, 0, -1)))
        res[name] = np.ascontiguousarray(a)
    return res


def read_Y(path):
    shp = _datasets(path)["Y"]
    return np.ascontiguousarray(_raw(path, "Y", "<f8").reshape(shp).T)   # julia (L, M)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference data not present; fixtures are committed, nothing to do")
    # ---- basic vbmf ----
    Y = read_Y(f"{REF}/vbmf_test/inputs.jld")
    log = read_log(f"{REF}/vbmf_test/log.jld")
    log.pop("YHat")
    np.savez_compressed(f"{OUT}/vbmf_test.npz", Y=Y, **log)
    print("vbmf_test:", {k: v.shape for k, v in log.items()})
    # ---- sparse ----
    Ys = read_Y(f"{REF}/sparse_test/inputs.jld")
    assert np.array_equal(Y, Ys)
    slog = read_log(f"{REF}/sparse_test/log.jld")
    slog.pop("YHat")
    keep = np.array([0, 1, 2, 100])
    slog["cov_slices"] = keep
    slog["SigmaATVec"] = slog["SigmaATVec"][keep]
    slog["invSigmaATVec"] = slog["invSigmaATVec"][keep]
    np.savez_compressed(f"{OUT}/sparse_test.npz", Y=Ys, **slog)
    print("sparse_test:", {k: v.shape for k, v in slog.items()})


if __name__ == "__main__":
    main()
