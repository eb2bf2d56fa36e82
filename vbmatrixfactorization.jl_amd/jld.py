"""JLD (HDF5) container of the reference's trajectory log -- src/data_manip.jl:53-93 (`save(...)` / `load(...)` of JLD.jl).

The reference persists its log with the JLD package: an HDF5 file with a 512-byte user block that starts with the
magic string "Julia data file (HDF5), version 0.1.1", one root dataset per dictionary key, a `/_creator` group, and
for non-plain values (`priors = Dict()`, src/vbmf.jl:177) JLD's `_refs` / `_types` machinery.  This module reads and
writes exactly that layout through ctypes on the image's libhdf5 (there is no h5py and nothing can be installed):

  * arrays: HDF5 dims are Julia's dims REVERSED and the bytes are Julia's column-major memory, so a Julia array of
    shape (d1, ..., dn) -- here a NumPy array of that shape -- is stored as `a.T` in C order with dims (dn, ..., d1);
  * `Int64` / `Float64` vectors (the log's scalar fields collected over time) are 1-D datasets;
  * empty arrays (`labels = Int64[]`) are NULL-dataspace datasets with an Int64 attribute "dims";
  * an empty `Dict()` is a scalar dataset of the committed compound `/_types/00000001` {keys_, values_} (attribute
    "julia type" = JLD.AssociativeWrapper{Core.Any,Core.Any,Base.Dict{Core.Any,Core.Any}}) holding object references to
    two NULL reference datasets `/_refs/0000000{1,2}` (attribute "julia eltype" = "Core.Any").

The layout was taken from the reference's own recorded files (examples/data/vbmf_test/{log,inputs}.jld, written by
Julia 0.5.2 / JLD) with `h5dump`; tests/test_jld.py re-saves that log and compares the two files' dumps object by object.
Julia itself is not available in this pipeline, so "the reference's load_log opens it" is verified structurally, not
by running Julia."""
import ctypes as C
import ctypes.util
import os

import numpy as np

MAGIC = b"Julia data file (HDF5), version 0.1.1"
USERBLOCK = 512
_CANDIDATES = ("/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so")

hid_t, herr_t, hsize_t = C.c_int64, C.c_int, C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0x0000, 0x0002
H5S_SCALAR, H5S_SIMPLE, H5S_NULL = 0, 1, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND, H5T_REFERENCE = 0, 1, 3, 6, 7
H5T_CSET_UTF8 = 1
H5R_OBJECT = 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5P_DEFAULT = 0

_lib = None


def lib():
    """libhdf5 (1.10 API, hid_t = int64).  VBMF_HDF5_LIB overrides the search."""
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get("VBMF_HDF5_LIB")] + list(_CANDIDATES) + [ctypes.util.find_library("hdf5")]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            L = C.CDLL(c)
            break
        except OSError as e:
            err = e
    else:
        raise ImportError(f"libhdf5 not found (tried {[c for c in cands if c]}): {err}; set VBMF_HDF5_LIB")
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    L.H5open()
    L.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise ImportError(f"libhdf5 {maj.value}.{mnr.value}.{rel.value}: the 1.10+ API (64-bit hid_t) is required")
    sig = {
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]), "H5Pcreate": (hid_t, [hid_t]), "H5Pset_userblock": (herr_t, [hid_t, hsize_t]),
        "H5Pget_userblock": (herr_t, [hid_t, C.POINTER(hsize_t)]), "H5Fget_create_plist": (hid_t, [hid_t]),
        "H5Pclose": (herr_t, [hid_t]), "H5Screate": (hid_t, [C.c_int]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_type": (C.c_int, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (herr_t, [hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]), "H5Tset_cset": (herr_t, [hid_t, C.c_int]),
        "H5Tcreate": (hid_t, [C.c_int, C.c_size_t]), "H5Tinsert": (herr_t, [hid_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Tcommit2": (herr_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Tclose": (herr_t, [hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (herr_t, [hid_t]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]),
        "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
        "H5Aclose": (herr_t, [hid_t]), "H5Aexists": (C.c_int, [hid_t, C.c_char_p]), "H5Aget_space": (hid_t, [hid_t]),
        "H5Aget_type": (hid_t, [hid_t]),
        "H5Rcreate": (herr_t, [C.c_void_p, hid_t, C.c_char_p, C.c_int, hid_t]),
        "H5Rdereference2": (hid_t, [hid_t, hid_t, C.c_int, C.c_void_p]),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    L.H5Eset_auto2(0, None, None)               # errors come back as negative ids / return codes, raised below
    _lib = L
    return L


_ITER_CB = C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)


def _g(name):
    """a predefined identifier (global hid_t of the library, valid after H5open)"""
    return hid_t.in_dll(lib(), name).value


def _chk(v, what):
    if v < 0:
        raise OSError(f"HDF5: {what} failed")
    return v


def _root_names(fid):
    L = lib()
    L.H5Literate.restype = herr_t
    L.H5Literate.argtypes = [hid_t, C.c_int, C.c_int, C.POINTER(hsize_t), _ITER_CB, C.c_void_p]
    names = []

    def cb(group, name, info, data):
        names.append(name.decode())
        return 0
    idx = hsize_t(0)
    _chk(L.H5Literate(fid, H5_INDEX_NAME, H5_ITER_INC, C.byref(idx), _ITER_CB(cb), None), "H5Literate")
    return names


def _read_numeric(did, what):
    L = lib()
    sid, tid = _chk(L.H5Dget_space(did), "H5Dget_space"), _chk(L.H5Dget_type(did), "H5Dget_type")
    try:
        cls, size, kind = L.H5Tget_class(tid), L.H5Tget_size(tid), L.H5Sget_simple_extent_type(sid)
        if cls == H5T_FLOAT and size == 8:
            dt, mem = np.float64, _g("H5T_NATIVE_DOUBLE_g")
        elif cls == H5T_INTEGER and size == 8:
            dt, mem = np.int64, _g("H5T_NATIVE_INT64_g")
        elif cls == H5T_INTEGER and size == 4:
            dt, mem = np.uint32, _g("H5T_NATIVE_UINT32_g")
        else:
            raise NotImplementedError(f"{what}: HDF5 class {cls} of {size} bytes is not a type the log holds")
        if kind == H5S_NULL:                                     # JLD's empty array: dims live in an attribute
            dims = ()
            if L.H5Aexists(did, b"dims") > 0:
                aid = _chk(L.H5Aopen(did, b"dims", H5P_DEFAULT), "H5Aopen")
                asp = L.H5Aget_space(aid)
                n = hsize_t(0)
                L.H5Sget_simple_extent_dims(asp, C.byref(n), None)
                d = np.zeros(int(n.value), dtype=np.int64)
                L.H5Aread(aid, _g("H5T_NATIVE_INT64_g"), d.ctypes.data_as(C.c_void_p))
                L.H5Sclose(asp); L.H5Aclose(aid)
                dims = tuple(int(x) for x in d)
            return np.empty(dims if dims else (0,), dtype=dt)
        if kind == H5S_SCALAR:
            out = np.zeros((), dtype=dt)
            _chk(L.H5Dread(did, mem, 0, 0, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), "H5Dread")
            return out[()]
        nd = L.H5Sget_simple_extent_ndims(sid)
        hd = (hsize_t * nd)()
        L.H5Sget_simple_extent_dims(sid, hd, None)
        h5shape = tuple(int(x) for x in hd)
        raw = np.empty(h5shape, dtype=dt)
        _chk(L.H5Dread(did, mem, 0, 0, H5P_DEFAULT, raw.ctypes.data_as(C.c_void_p)), "H5Dread")
        return np.ascontiguousarray(raw.T)                       # Julia's dims and values: reversed dims, column-major bytes
    finally:
        L.H5Tclose(tid); L.H5Sclose(sid)


def _read_dict(fid, did, what):
    """JLD's AssociativeWrapper: only the empty Dict the reference writes (src/vbmf.jl:177) is decoded."""
    L = lib()
    refs = (C.c_uint64 * 2)()
    tid = L.H5Dget_type(did)
    try:
        _chk(L.H5Dread(did, tid, 0, 0, H5P_DEFAULT, C.cast(refs, C.c_void_p)), "H5Dread")
    finally:
        L.H5Tclose(tid)
    for k in range(2):
        one = C.c_uint64(refs[k])
        oid = _chk(L.H5Rdereference2(fid, H5P_DEFAULT, H5R_OBJECT, C.byref(one)), "H5Rdereference2")
        sid = L.H5Dget_space(oid)
        kind = L.H5Sget_simple_extent_type(sid)
        L.H5Sclose(sid); L.H5Dclose(oid)
        if kind != H5S_NULL:
            raise NotImplementedError(f"{what}: a non-empty Dict is not a value the reference's log writer produces")
    return {}


def load(path):
    """JLD.load(path): {name: value} of the root datasets (`_creator`, `_refs`, `_types` are JLD's own)."""
    L = lib()
    with open(path, "rb") as f:
        head = f.read(len(MAGIC))
    if not head.startswith(b"Julia data file (HDF5)"):
        raise OSError(f"{path}: no JLD magic in the user block")
    fid = _chk(L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT), f"H5Fopen({path})")
    out = {}
    try:
        for name in _root_names(fid):
            if name.startswith("_"):
                continue
            did = L.H5Dopen2(fid, name.encode(), H5P_DEFAULT)
            if did < 0:
                continue                                         # a group of the caller's own: not part of a log
            try:
                tid = L.H5Dget_type(did)
                cls = L.H5Tget_class(tid)
                L.H5Tclose(tid)
                out[name] = _read_dict(fid, did, name) if cls == H5T_COMPOUND else _read_numeric(did, name)
            finally:
                L.H5Dclose(did)
    finally:
        L.H5Fclose(fid)
    return out


def creator(path):
    """the `/_creator` record of a JLD file (Julia version, word size, byte-order mark)"""
    L = lib()
    fid = _chk(L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT), f"H5Fopen({path})")
    try:
        out = {}
        for k in ("ENDIAN_BOM", "JULIA_MAJOR", "JULIA_MINOR", "JULIA_PATCH", "WORD_SIZE"):
            did = _chk(L.H5Dopen2(fid, f"/_creator/{k}".encode(), H5P_DEFAULT), f"/_creator/{k}")
            try:
                out[k] = int(_read_numeric(did, k))
            finally:
                L.H5Dclose(did)
        return out
    finally:
        L.H5Fclose(fid)


def _write_array(fid, name, a):
    L = lib()
    a = np.asarray(a)
    if a.dtype.kind in "iub":
        a, mem, ftype = a.astype(np.int64), _g("H5T_NATIVE_INT64_g"), _g("H5T_STD_I64LE_g")
    elif a.dtype.kind == "f":
        a, mem, ftype = a.astype(np.float64), _g("H5T_NATIVE_DOUBLE_g"), _g("H5T_IEEE_F64LE_g")
    else:
        raise TypeError(f"{name}: dtype {a.dtype} is not a type the log holds")
    if a.ndim == 0:
        a = a.reshape(1)
    if a.size == 0:                                              # JLD: NULL dataspace + "dims"
        sid = _chk(L.H5Screate(H5S_NULL), "H5Screate")
        did = _chk(L.H5Dcreate2(fid, name.encode(), ftype, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({name})")
        dims = np.array(a.shape, dtype=np.int64)
        n = (hsize_t * 1)(dims.size)
        asp = L.H5Screate_simple(1, n, None)
        aid = _chk(L.H5Acreate2(did, b"dims", _g("H5T_STD_I64LE_g"), asp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2(dims)")
        L.H5Awrite(aid, _g("H5T_NATIVE_INT64_g"), dims.ctypes.data_as(C.c_void_p))
        L.H5Aclose(aid); L.H5Sclose(asp); L.H5Dclose(did); L.H5Sclose(sid)
        return
    raw = np.ascontiguousarray(a.T)                              # reversed dims, Julia's column-major bytes
    hd = (hsize_t * raw.ndim)(*raw.shape)
    sid = _chk(L.H5Screate_simple(raw.ndim, hd, None), "H5Screate_simple")
    did = _chk(L.H5Dcreate2(fid, name.encode(), ftype, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({name})")
    try:
        _chk(L.H5Dwrite(did, mem, 0, 0, H5P_DEFAULT, raw.ctypes.data_as(C.c_void_p)), f"H5Dwrite({name})")
    finally:
        L.H5Dclose(did); L.H5Sclose(sid)


def _write_scalar(loc, name, value, ftype_g, mem_g, ctype):
    L = lib()
    sid = L.H5Screate(H5S_SCALAR)
    did = _chk(L.H5Dcreate2(loc, name.encode(), _g(ftype_g), sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({name})")
    v = ctype(value)
    L.H5Dwrite(did, _g(mem_g), 0, 0, H5P_DEFAULT, C.byref(v))
    L.H5Dclose(did); L.H5Sclose(sid)


def _string_attr(obj, name, text):
    L = lib()
    b = text.encode()
    t = L.H5Tcopy(_g("H5T_C_S1_g"))
    L.H5Tset_size(t, len(b))                                     # JLD sizes the string exactly (no room for the terminator)
    L.H5Tset_cset(t, H5T_CSET_UTF8)
    sid = L.H5Screate(H5S_SCALAR)
    aid = _chk(L.H5Acreate2(obj, name.encode(), t, sid, H5P_DEFAULT, H5P_DEFAULT), f"H5Acreate2({name})")
    buf = C.create_string_buffer(b, len(b))
    L.H5Awrite(aid, t, buf)
    L.H5Aclose(aid); L.H5Sclose(sid); L.H5Tclose(t)


def _write_empty_dict(fid, name, state):
    """`Dict()` as JLD writes it (see the module docstring); `state` numbers the `_refs` / `_types` entries per file."""
    L = lib()
    if "types" not in state:
        state["refs_g"] = _chk(L.H5Gcreate2(fid, b"_refs", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2(_refs)")
        state["types_g"] = _chk(L.H5Gcreate2(fid, b"_types", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2(_types)")
        ref_t = _g("H5T_STD_REF_OBJ_g")
        ct = _chk(L.H5Tcreate(H5T_COMPOUND, 16), "H5Tcreate")
        L.H5Tinsert(ct, b"keys_", 0, ref_t)
        L.H5Tinsert(ct, b"values_", 8, ref_t)
        _chk(L.H5Tcommit2(fid, b"/_types/00000001", ct, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Tcommit2")
        _string_attr(ct, "julia type", "JLD.AssociativeWrapper{Core.Any,Core.Any,Base.Dict{Core.Any,Core.Any}}")
        state["types"] = ct
        state["nref"] = 0
    refs = (C.c_uint64 * 2)()
    for k in range(2):
        state["nref"] += 1
        rname = f"/_refs/{state['nref']:08d}"
        sid = L.H5Screate(H5S_NULL)
        did = _chk(L.H5Dcreate2(fid, rname.encode(), _g("H5T_STD_REF_OBJ_g"), sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), rname)
        _string_attr(did, "julia eltype", "Core.Any")
        L.H5Dclose(did); L.H5Sclose(sid)
        one = C.c_uint64(0)
        _chk(L.H5Rcreate(C.byref(one), fid, rname.encode(), H5R_OBJECT, -1), "H5Rcreate")
        refs[k] = one.value
    sid = L.H5Screate(H5S_SCALAR)
    did = _chk(L.H5Dcreate2(fid, name.encode(), state["types"], sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({name})")
    L.H5Dwrite(did, state["types"], 0, 0, H5P_DEFAULT, C.cast(refs, C.c_void_p))
    L.H5Dclose(did); L.H5Sclose(sid)


def save(path, data, creator_version=(0, 5, 2)):
    """JLD.save(path, dict): one root dataset per key.  `creator_version` fills `/_creator` (the reference's files were
    written by Julia 0.5.2, the version its REQUIRE names); values: numeric arrays / scalars, or an empty dict."""
    L = lib()
    fcpl = _chk(L.H5Pcreate(_g("H5P_CLS_FILE_CREATE_ID_g")), "H5Pcreate")
    _chk(L.H5Pset_userblock(fcpl, USERBLOCK), "H5Pset_userblock")
    fid = _chk(L.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, fcpl, H5P_DEFAULT), f"H5Fcreate({path})")
    L.H5Pclose(fcpl)
    state = {}
    try:
        for name, v in data.items():
            if isinstance(v, dict):
                if v:
                    raise NotImplementedError(f"{name}: only the empty Dict() the reference writes is encoded")
                _write_empty_dict(fid, name, state)
            else:
                _write_array(fid, name, v)
        g = _chk(L.H5Gcreate2(fid, b"_creator", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2(_creator)")
        _write_scalar(g, "ENDIAN_BOM", 0x04030201, "H5T_STD_U32LE_g", "H5T_NATIVE_UINT32_g", C.c_uint32)
        for k, val in zip(("JULIA_MAJOR", "JULIA_MINOR", "JULIA_PATCH"), creator_version):
            _write_scalar(g, k, val, "H5T_STD_I64LE_g", "H5T_NATIVE_INT64_g", C.c_int64)
        _write_scalar(g, "WORD_SIZE", 64, "H5T_STD_I64LE_g", "H5T_NATIVE_INT64_g", C.c_int64)
        L.H5Gclose(g)
    finally:
        if "types" in state:
            L.H5Tclose(state["types"]); L.H5Gclose(state["refs_g"]); L.H5Gclose(state["types_g"])
        L.H5Fclose(fid)
    with open(path, "r+b") as f:                                 # JLD's magic at the start of the user block
        f.write(MAGIC + b"\x00" * (USERBLOCK - len(MAGIC)))
