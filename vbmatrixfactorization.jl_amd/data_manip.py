"""Trajectory log writer/reader -- the twin of src/data_manip.jl:6-118 (create_log, update_log!, save_log,
load_log, extract_params!), SURVEY.md section 8f row N2.

Same structure as the reference's log: a dictionary with one entry per field of the parameter type, scalars
collected into vectors, arrays concatenated along a NEW TRAILING axis (time is always the last dimension,
src/data_manip.jl:41,97-99); slice 0 is the state before the first sweep (src/vbmf.jl:181-184).
On disk it is the reference's own container: JLD (HDF5) files  <logdir>/<desc>/log.jld  and  inputs.jld
(src/data_manip.jl:62-68), written and read by jld.py through the image's libhdf5 with the layout of the files the
reference itself recorded -- so `load_log` opens the reference's examples/data/*/ directories directly and the
reference's tooling sees its own format.  Fields that are None (YHat beyond the materialisation limit, the dense
MH x MH covariances of the full_cov branch) are not logged.  Indices are 0-based (the reference's t is 1-based)."""
import dataclasses
import datetime
import json
import os

import numpy as np

from . import jld


def _named_values(params):
    for f in dataclasses.fields(params):
        yield f.name, getattr(params, f.name)


def _is_scalar(v):
    return isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool)


def create_log(params):
    """create_log -- src/data_manip.jl:6-25."""
    logVar = {}
    for name, v in _named_values(params):
        if v is None:
            continue
        logVar[name] = np.array([v]) if _is_scalar(v) else np.array(v, copy=True)
    return logVar


def update_log_(logVar, params):
    """update_log! -- src/data_manip.jl:32-45: cat(1, ...) for scalars, cat(ndims+1, ...) for arrays."""
    for name, v in _named_values(params):
        if v is None or name not in logVar:
            continue
        if _is_scalar(v):
            logVar[name] = np.concatenate([logVar[name], np.array([v])])
        else:
            v = np.asarray(v)
            old = logVar[name]
            if old.ndim == v.ndim:                       # first update: the log holds the bare initial array
                old = old[..., None]
            logVar[name] = np.concatenate([old, v[..., None]], axis=-1)
    return logVar


def save_log(logVar, Y, priors, logdir, desc=""):
    """save_log -- src/data_manip.jl:53-69: <logdir>/<desc>/log.jld and inputs.jld (JLD = HDF5 with Julia's conventions:
    dims reversed, column-major bytes; see jld.py).  `priors` is the reference's `Dict()` (src/vbmf.jl:177): always empty.
    Returns the directory written."""
    if desc == "":
        desc = datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
    d = os.path.join(logdir, desc)
    os.makedirs(d, exist_ok=True)
    jld.save(os.path.join(d, "log.jld"), logVar)
    jld.save(os.path.join(d, "inputs.jld"), {"Y": np.asarray(Y, dtype=np.float64), "priors": dict(priors or {})})
    return d


def load_log(path):
    """load_log -- src/data_manip.jl:77-93: (logVar, Y, priors) from <path>/log.jld and <path>/inputs.jld -- e.g. the
    reference's own examples/data/vbmf_test.  (A directory written by round 1 of this build, log.npz / inputs.npz, is
    still readable.)"""
    try:
        if os.path.exists(os.path.join(path, "log.jld")):
            logVar = jld.load(os.path.join(path, "log.jld"))
            inputs = jld.load(os.path.join(path, "inputs.jld"))
            return logVar, inputs["Y"], inputs["priors"]
        with np.load(os.path.join(path, "log.npz")) as z:
            logVar = {k: z[k] for k in z.files}
        with np.load(os.path.join(path, "inputs.npz")) as z:
            Y, priors = z["Y"], json.loads(str(z["priors"]))
        return logVar, Y, priors
    except (OSError, KeyError) as e:
        ls = os.listdir(path) if os.path.isdir(path) else []
        raise RuntimeError("The specified folder does not contain any log files but it contains the following:\n"
                           + "\n".join(ls)) from e


def nslices(logVar):
    """Number of time slices in a log (length of any scalar field's vector)."""
    for v in logVar.values():
        if v.ndim == 1:
            return v.shape[0]
    raise ValueError("log without scalar fields")


def extract_params_(logVar, t, params):
    """extract_params! -- src/data_manip.jl:100-118: slice t (0-based; time is the last axis) into `params`."""
    for name, _ in _named_values(params):
        if name not in logVar:
            continue
        a = logVar[name]
        if a.ndim == 1:
            v = a[t]
            setattr(params, name, int(v) if np.issubdtype(a.dtype, np.integer) else float(v))
        else:
            setattr(params, name, np.array(a[..., t], copy=True))
    return params
