"""The Julia host (vbmatrixfactorization.jl_amd/julia/VBMatrixFactorizationHIP.jl) cannot be executed in this pipeline (no
`julia` binary), so its `ccall`s are checked STATICALLY against include/vbmf_hip.h: every called symbol is declared there,
with the same number of arguments, and every argument / return type maps onto the C type (Int64 <-> int64_t, Cint <-> int,
Float64 <-> double, Ptr{Float64} / Ref{Float64} <-> double*, Ptr{Cvoid} <-> the opaque handles, ...); the two structs
passed by reference mirror the header's field lists.  A drifted signature fails here instead of shipping silently."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vbmf_hip.h")
JULIA = os.path.join(ROOT, "vbmatrixfactorization.jl_amd", "julia", "VBMatrixFactorizationHIP.jl")

# C parameter type (normalised: no const, no spaces before *) -> Julia types a ccall may name for it
C2J = {
    "int": {"Cint", "Int32"}, "int32_t": {"Cint", "Int32"}, "int64_t": {"Int64", "Clonglong"}, "uint64_t": {"UInt64"},
    "double": {"Float64", "Cdouble"}, "size_t": {"Csize_t", "UInt"},
    "double*": {"Ptr{Float64}", "Ref{Float64}"}, "int64_t*": {"Ptr{Int64}", "Ref{Int64}"}, "uint32_t*": {"Ptr{UInt32}"},
    "void*": {"Ptr{Cvoid}"}, "vbmf_ctx*": {"Ptr{Cvoid}"}, "vbmf_prep*": {"Ptr{Cvoid}"},
    "vbmf_ctx**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"}, "vbmf_prep**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "vbmf_opts*": {"Ref{VbmfOpts}", "Ptr{VbmfOpts}"}, "vbmf_sparse_hyper*": {"Ref{SparseHyper}", "Ptr{SparseHyper}"},
    "char*": {"Cstring"},
}
FIELD2J = {"int32_t": "Int32", "uint32_t": "UInt32", "int64_t": "Int64", "double": "Float64"}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _ctype(decl):
    """'const double* AHat' -> 'double*' (the parameter name, if any, is dropped)"""
    d = decl.replace("const", " ").strip()
    stars = d.count("*")
    words = d.replace("*", " ").split()
    base = words[0] if words[0] not in ("unsigned", "struct") else " ".join(words[:2])
    return base + "*" * stars


def header_prototypes():
    text = _strip_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*(const\s+char\s*\*|int|void)\s+(vbmf_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = ("char*" if "char" in ret else ret, params)
    return protos


def header_struct(name):
    text = _strip_comments(open(HEADER).read())
    m = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*" + name + r"\s*;", text, flags=re.S)
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        fields += [(n.strip(), ty) for n in names.split(",")]
    return fields


def julia_ccalls():
    src = open(JULIA).read()
    src = "\n".join(l.split("#", 1)[0] if not re.search(r'"[^"]*#', l) else l for l in src.split("\n"))
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*libvbmf\),\s*([\w{}]+),\s*\(([^)]*)\)\s*,", src, flags=re.S):
        types = [t.strip() for t in re.split(r",(?![^{]*\})", m.group(3)) if t.strip()]
        # the actual arguments: up to the parenthesis that closes the ccall
        i, depth, start = m.end(), 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = [a for a in re.split(r",(?![^\[\(]*[\]\)])", src[start:i - 1]) if a.strip()]
        calls.append((m.group(1), m.group(2), types, len(args), src.count("\n", 0, m.start()) + 1))
    return calls


def julia_struct(name):
    src = open(JULIA).read()
    m = re.search(r"struct\s+" + name + r"\b(.*?)\nend", src, flags=re.S)
    body = re.sub(r"#[^\n]*", "", m.group(1))
    return [(a.strip(), b.strip()) for a, b in re.findall(r"(\w+)\s*::\s*(\w+)", body)]


def test_every_ccall_matches_the_header():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(protos) >= 47 and len(calls) >= 40, (len(protos), len(calls))
    problems = []
    for sym, ret, types, nargs, line in calls:
        if sym not in protos:
            problems.append(f"line {line}: {sym} is not declared in include/vbmf_hip.h")
            continue
        cret, cparams = protos[sym]
        if ret not in C2J[cret]:
            problems.append(f"line {line}: {sym} returns {cret}, ccall says {ret}")
        if len(types) != len(cparams) or nargs != len(cparams):
            problems.append(f"line {line}: {sym} takes {len(cparams)} arguments, ccall names {len(types)} types and passes {nargs} values")
            continue
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            if jt not in C2J.get(ct, ()):
                problems.append(f"line {line}: {sym} argument {k + 1} is {ct}, ccall says {jt}")
    assert not problems, "\n".join(problems)


def test_the_bound_surface_covers_the_reference_entry_points():
    called = {c[0] for c in julia_ccalls()}
    need = {"vbmf_create", "vbmf_destroy", "vbmf_last_error", "vbmf_set_Y", "vbmf_set_state", "vbmf_get_state", "vbmf_step",
            "vbmf_run", "vbmf_run_fixed_basis", "vbmf_get_YHat", "vbmf_sparse_set_state", "vbmf_sparse_get_state", "vbmf_sparse_run",
            "vbmf_sparse_step", "vbmf_sparse_lower_bound", "vbmf_sparse_lower_bound_trimmed", "vbmf_sparse_set_noise_rows",
            "vbmf_sparse_get_noise_rows", "vbmf_sparse_set_full_cov", "vbmf_sparse_set_SigmaA", "vbmf_sparse_get_SigmaA",
            "vbmf_dual_set_priors", "vbmf_dual_get_priors", "vbmf_dual_run",
            "vbmf_trial_set_priors", "vbmf_trial_get_priors", "vbmf_trial_run",
            "vbmf_preprocess_open", "vbmf_preprocess_rows", "vbmf_set_Y_preprocessed", "vbmf_preprocess_close"}
    assert need <= called, sorted(need - called)


@pytest.mark.parametrize("cname,jname", [("vbmf_opts", "VbmfOpts"), ("vbmf_sparse_hyper", "SparseHyper")])
def test_structs_passed_by_reference_mirror_the_header(cname, jname):
    want = [(n, FIELD2J[t]) for n, t in header_struct(cname)]
    assert julia_struct(jname) == want


def test_parameter_structs_keep_the_reference_field_order():
    """field names and order of the reference's structs (src/vbmf.jl:22-40) in both hosts"""
    import dataclasses
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    pkg = G.load_package()
    ref_basic = ["L", "M", "H", "H1", "labels", "AHat", "BHat", "SigmaA", "SigmaB", "CA", "CB", "invCA", "invCB", "sigma2", "YHat"]
    assert [f.name for f in dataclasses.fields(pkg.vbmf_parameters)] == ref_basic
    assert [n for n, _ in julia_struct("vbmf_parameters")] == ref_basic
    # the other three Julia structs are the Python ones minus the dense MH x MH pair
    for jn, cls in (("vbmf_sparse_parameters", pkg.vbmf_sparse_parameters), ("vbmf_dual_parameters", pkg.vbmf_dual_parameters),
                    ("vbmf_trial_parameters", pkg.vbmf_trial_parameters)):
        py = [f.name for f in dataclasses.fields(cls) if f.name not in ("SigmaATVec", "invSigmaATVec")]
        assert [n for n, _ in julia_struct(jn)] == py, jn
