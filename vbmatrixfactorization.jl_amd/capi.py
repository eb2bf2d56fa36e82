"""ctypes binding of include/vbmf_hip.h -- the same entry points a Julia `ccall` binds.

The library is loaded from this directory only (built in-tree by build.py).  If it is missing the
import fails loudly: there is no CPU fallback and no alternate backend.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VBMF_HIP_LIB: another build of the SAME library (A/B tuning builds under variants/); default: the in-tree build.
# An override must exist and export every symbol of include/vbmf_hip.h (lib() binds them all and fails loudly otherwise); it is NOT
# covered by build.py's source-hash stamp -- whoever builds a variant is responsible for building it from the current sources.
LIB_PATH = os.environ.get("VBMF_HIP_LIB") or os.path.join(_HERE, "libvbmf_hip.so")
if os.environ.get("VBMF_HIP_LIB") and not os.path.isfile(LIB_PATH):
    raise FileNotFoundError(f"VBMF_HIP_LIB={LIB_PATH}: no such library (A/B variant builds: scripts/README.md)")

VBMF_Y_F32, VBMF_Y_BF16 = 0, 1
VBMF_FACTOR_AUTO, VBMF_FACTOR_BF16, VBMF_FACTOR_BF16X2 = 0, 1, 2
VBMF_FACTOR_BF16_MAX_H = 128      # vbmf_create refuses the single-bf16 factor operand above this rank (include/vbmf_hip.h)
VBMF_VARIANT_BASIC, VBMF_VARIANT_SPARSE_DIAG, VBMF_VARIANT_SPARSE_DIAGVAR, VBMF_VARIANT_DUAL_DIAG, VBMF_VARIANT_TRIAL_DIAG = 0, 1, 2, 3, 4
VBMF_VARIANT_DUAL_DIAGVAR, VBMF_VARIANT_TRIAL_DIAGVAR = 5, 6
VBMF_COMPAT_SPECTRAL_DELTA, VBMF_COMPAT_SPARSE_REPEAT, VBMF_COMPAT_DEFAULT = 1, 2, 0xFFFFFFFF
STEP_A, STEP_B, STEP_CA, STEP_CB, STEP_SIGMA2 = 1, 2, 4, 8, 16
UNIQUE_ID_BYTES = 128

# every symbol include/vbmf_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "vbmf_default_opts", "vbmf_create", "vbmf_destroy", "vbmf_last_error", "vbmf_set_Y", "vbmf_set_Y_synthetic",
    "vbmf_get_Y", "vbmf_get_trYY", "vbmf_set_state", "vbmf_get_state", "vbmf_step", "vbmf_run", "vbmf_run_fixed_basis",
    "vbmf_get_YHat",
    "vbmf_elbo", "vbmf_comm_unique_id", "vbmf_comm_init", "vbmf_comm_set_transport", "vbmf_profile_enable", "vbmf_profile_read",
    "vbmf_pass_bytes", "vbmf_device_sync", "vbmf_debug_peek", "vbmf_debug_time_pass", "vbmf_debug_lambda_max",
    "vbmf_sparse_set_state", "vbmf_sparse_get_state", "vbmf_sparse_step", "vbmf_sparse_run", "vbmf_sparse_run_fixed_basis",
    "vbmf_sparse_lower_bound", "vbmf_sparse_set_noise_rows", "vbmf_sparse_get_noise_rows", "vbmf_preprocess_open", "vbmf_preprocess_rows", "vbmf_set_Y_preprocessed",
    "vbmf_preprocess_close", "vbmf_dual_set_priors", "vbmf_dual_get_priors", "vbmf_dual_run",
    "vbmf_trial_set_priors", "vbmf_trial_get_priors", "vbmf_trial_run",
    "vbmf_sparse_set_full_cov", "vbmf_sparse_set_SigmaA", "vbmf_sparse_get_SigmaA",
    "vbmf_sparse_lower_bound_trimmed", "vbmf_debug_set",
]
VBMF_OK, VBMF_ERR_INVALID, VBMF_ERR_NO_DEVICE, VBMF_ERR_HIP, VBMF_ERR_NUMERIC, VBMF_ERR_COMM, VBMF_ERR_UNSUPPORTED, VBMF_ERR_SYNC = 0, -1, -2, -3, -4, -5, -6, -7
DEBUG_EPI_SPIN_LIMIT, DEBUG_EPI_EXPECT_SKEW, DEBUG_SIGMA_B_PPM, DEBUG_EXACT_LAMBDA = 0, 1, 2, 3
SSTEP_A, SSTEP_B, SSTEP_CA, SSTEP_CB, SSTEP_SIGMA, SSTEP_PRIORS = 1, 2, 4, 8, 16, 32
PEEK_P, PEEK_Q, PEEK_A32, PEEK_B32, PEEK_FA, PEEK_FB, PEEK_Y1, PEEK_Y2, PEEK_DIMS, PEEK_CHAIN = range(10)


class VbmfOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("y_dtype", C.c_int32), ("factor_dtype", C.c_int32),
        ("variant", C.c_int32), ("reference_compat", C.c_uint32), ("nranks", C.c_int32), ("rank", C.c_int32),
        ("L_global", C.c_int64), ("row_offset", C.c_int64), ("pass1_splits", C.c_int32), ("reserved", C.c_int32),
    ]


class VbmfSparseHyper(C.Structure):
    _fields_ = [("alpha0", C.c_double), ("beta0", C.c_double), ("gamma0", C.c_double), ("delta0", C.c_double),
                ("eta0", C.c_double), ("zeta0", C.c_double)]


# int fn(void* user, void* buf, size_t count, int is_double, void* hip_stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class VbmfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vbmf_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libvbmf_hip.so (once).  Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # Load order matters in a process that also uses PyTorch-ROCm: torch bundles its own copies of
    # libamdhip64/librccl (same SONAMEs as /opt/rocm's).  If torch is imported AFTER this library the
    # process aborts at interpreter exit (double free in the duplicated runtime); torch first is fine --
    # this library then binds to the runtime torch loaded.  So load torch (when present) first.  It is
    # not used for anything here.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python vbmatrixfactorization.jl_amd/build.py` "
            "(hipcc, gfx950).  This package has no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    dp, i64, i32, vp = C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_void_p
    L.vbmf_default_opts.argtypes = [C.POINTER(VbmfOpts)]
    L.vbmf_default_opts.restype = None
    L.vbmf_create.argtypes = [C.POINTER(vp), i64, i64, i64, C.POINTER(VbmfOpts)]
    L.vbmf_destroy.argtypes = [vp]
    L.vbmf_last_error.argtypes = [vp]
    L.vbmf_last_error.restype = C.c_char_p
    L.vbmf_set_Y.argtypes = [vp, dp, i64]
    L.vbmf_set_Y_synthetic.argtypes = [vp, C.c_uint64, i64, C.c_double]
    L.vbmf_get_Y.argtypes = [vp, dp, i64, i64, i64]
    L.vbmf_get_trYY.argtypes = [vp, dp]
    L.vbmf_set_state.argtypes = [vp, dp, i64, dp, i64, dp, dp, dp, dp, C.c_double, C.POINTER(i64), i64, i64]
    L.vbmf_get_state.argtypes = [vp, dp, i64, dp, i64, dp, dp, dp, dp, dp]
    L.vbmf_step.argtypes = [vp, i32]
    L.vbmf_run.argtypes = [vp, i64, C.c_double, i32, i32, C.POINTER(i64), dp, dp]
    L.vbmf_run_fixed_basis.argtypes = [vp, i64]
    L.vbmf_sparse_run_fixed_basis.argtypes = [vp, i64]
    L.vbmf_get_YHat.argtypes = [vp, dp, i64]
    L.vbmf_elbo.argtypes = [vp, dp]
    L.vbmf_comm_unique_id.argtypes = [vp]
    L.vbmf_comm_init.argtypes = [vp, vp]
    L.vbmf_comm_set_transport.argtypes = [vp, ALLREDUCE_FN, vp]
    L.vbmf_profile_enable.argtypes = [vp, i32]
    L.vbmf_profile_read.argtypes = [vp, dp, i32]
    L.vbmf_pass_bytes.argtypes = [vp, i32, dp]
    L.vbmf_device_sync.argtypes = [vp]
    L.vbmf_debug_peek.argtypes = [vp, i32, C.POINTER(C.c_uint32), i64, i64]
    L.vbmf_debug_time_pass.argtypes = [vp, i32, i32, dp]
    L.vbmf_debug_lambda_max.argtypes = [vp, dp, dp, dp]
    L.vbmf_sparse_set_state.argtypes = [vp, dp, dp, dp, dp, dp, i64, dp, dp, dp, C.c_double, C.c_double,
                                        C.POINTER(VbmfSparseHyper), C.POINTER(i64), i64, i64]
    L.vbmf_sparse_get_state.argtypes = [vp, dp, dp, dp, dp, dp, dp, i64, dp, dp, dp, dp, dp]
    L.vbmf_sparse_step.argtypes = [vp, i32]
    L.vbmf_sparse_run.argtypes = [vp, i64, C.c_double, i32, C.POINTER(i64), dp, dp]
    L.vbmf_sparse_lower_bound.argtypes = [vp, i32, dp]
    L.vbmf_sparse_lower_bound_trimmed.argtypes = [vp, i32, C.c_double, dp]
    L.vbmf_debug_set.argtypes = [vp, i32, i64]
    L.vbmf_sparse_set_full_cov.argtypes = [vp, i32]
    L.vbmf_sparse_set_SigmaA.argtypes = [vp, dp]
    L.vbmf_sparse_get_SigmaA.argtypes = [vp, dp]
    L.vbmf_dual_set_priors.argtypes = [vp, i64] + [C.c_double] * 6
    L.vbmf_dual_get_priors.argtypes = [vp, C.POINTER(i64), dp]
    L.vbmf_dual_run.argtypes = [vp, i64, C.c_double, i32, i32, C.POINTER(i64), dp, dp]
    L.vbmf_trial_set_priors.argtypes = [vp, i64, i64, dp]
    L.vbmf_trial_get_priors.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), dp]
    L.vbmf_trial_run.argtypes = [vp, i64, C.c_double, i32, i32, C.POINTER(i64), dp, dp]
    L.vbmf_sparse_set_noise_rows.argtypes = [vp, dp, dp, C.c_double]
    L.vbmf_sparse_get_noise_rows.argtypes = [vp, dp, dp]
    L.vbmf_preprocess_open.argtypes = [C.POINTER(vp), i32, dp, i64, i64, i64, C.POINTER(i64)]
    L.vbmf_preprocess_rows.argtypes = [vp, C.POINTER(i64), dp, dp]
    L.vbmf_set_Y_preprocessed.argtypes = [vp, vp, C.c_double]
    L.vbmf_preprocess_close.argtypes = [vp]
    for name in SYMBOLS:
        if name not in ("vbmf_default_opts", "vbmf_last_error"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _fcol(a, shape=None):
    """float64, column-major (Julia Array{Float64,2} memory)."""
    a = np.asarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return np.asfortranarray(a)


class PreprocessPlan:
    """preprocess (src/util.jl:73-86) fused into the upload: holds the caller's fp64 Y on the device with its row
    statistics and the kept rows; Context.set_Y_preprocessed tiles from it.  Use as a context manager."""

    def __init__(self, Y, device=0):
        Yf = np.asfortranarray(Y, dtype=np.float64)
        self.L, self.M = Yf.shape
        self._lib = lib()
        self._h = C.c_void_p()
        n = C.c_int64()
        rc = self._lib.vbmf_preprocess_open(C.byref(self._h), device, _dptr(Yf), self.L, self.M, self.L, C.byref(n))
        if rc != 0:
            raise VbmfError(rc, self._lib.vbmf_last_error(None).decode())
        self.L_used = n.value

    def rows(self):
        r = np.empty(self.L_used, dtype=np.int64); mu = np.empty(self.L); den = np.empty(self.L)
        self._lib.vbmf_preprocess_rows(self._h, r.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(mu), _dptr(den))
        return r, mu, den

    def close(self):
        if self._h:
            self._lib.vbmf_preprocess_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One problem on one GPU: thin, explicit wrapper over the C ABI (no numerics on the host)."""

    def __init__(self, L, M, H, y_dtype=VBMF_Y_BF16, factor_dtype=VBMF_FACTOR_AUTO, device=0, nranks=1, rank=0,
                 L_global=0, row_offset=0, reference_compat=VBMF_COMPAT_DEFAULT, pass1_splits=0,
                 variant=VBMF_VARIANT_BASIC):
        self._lib = lib()
        o = VbmfOpts()
        self._lib.vbmf_default_opts(C.byref(o))
        o.device, o.y_dtype, o.factor_dtype = device, y_dtype, factor_dtype
        o.variant = variant
        o.nranks, o.rank, o.L_global, o.row_offset = nranks, rank, L_global, row_offset
        o.reference_compat, o.pass1_splits = reference_compat, pass1_splits
        self._h = C.c_void_p()
        rc = self._lib.vbmf_create(C.byref(self._h), L, M, H, C.byref(o))
        if rc != 0:
            msg = self._lib.vbmf_last_error(None).decode()
            self._h = C.c_void_p()
            raise VbmfError(rc, msg)
        self.L, self.M, self.H = int(L), int(M), int(H)

    def _chk(self, rc):
        if rc != 0:
            raise VbmfError(rc, self._lib.vbmf_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.vbmf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- Y ----
    def set_Y(self, Y):
        Y = _fcol(Y, (self.L, self.M))
        self._chk(self._lib.vbmf_set_Y(self._h, _dptr(Y), Y.shape[0]))

    def set_Y_preprocessed(self, plan, lam):
        self._chk(self._lib.vbmf_set_Y_preprocessed(self._h, plan._h, float(lam)))

    def set_Y_synthetic(self, seed, Hstar, noise_std):
        self._chk(self._lib.vbmf_set_Y_synthetic(self._h, seed, Hstar, noise_std))

    def get_Y(self, row0=0, nrows=None):
        nrows = self.L - row0 if nrows is None else nrows
        out = np.empty((nrows, self.M), dtype=np.float64, order="F")
        self._chk(self._lib.vbmf_get_Y(self._h, _dptr(out), nrows, row0, nrows))
        return out

    def trYY(self):
        v = C.c_double()
        self._chk(self._lib.vbmf_get_trYY(self._h, C.byref(v)))
        return v.value

    # ---- state ----
    def set_state(self, AHat, BHat, SigmaA, SigmaB, CA_diag, CB_diag, sigma2, labels0=(), H1=0):
        A = _fcol(AHat, (self.M, self.H)); B = _fcol(BHat, (self.L, self.H))
        SA = _fcol(SigmaA, (self.H, self.H)); SB = _fcol(SigmaB, (self.H, self.H))
        ca = np.ascontiguousarray(CA_diag, dtype=np.float64); cb = np.ascontiguousarray(CB_diag, dtype=np.float64)
        if ca.shape != (self.H,) or cb.shape != (self.H,):
            raise ValueError("CA_diag/CB_diag must have length H")
        lab = np.ascontiguousarray(labels0, dtype=np.int64)
        self._chk(self._lib.vbmf_set_state(self._h, _dptr(A), self.M, _dptr(B), self.L, _dptr(SA), _dptr(SB), _dptr(ca),
                                           _dptr(cb), float(sigma2), lab.ctypes.data_as(C.POINTER(C.c_int64)),
                                           lab.size, int(H1)))

    def get_state(self, want_B=True):
        A = np.empty((self.M, self.H), order="F"); B = np.empty((self.L, self.H), order="F") if want_B else None
        SA = np.empty((self.H, self.H), order="F"); SB = np.empty((self.H, self.H), order="F")
        ca = np.empty(self.H); cb = np.empty(self.H); s2 = C.c_double()
        self._chk(self._lib.vbmf_get_state(self._h, _dptr(A), self.M, _dptr(B), self.L, _dptr(SA), _dptr(SB), _dptr(ca),
                                           _dptr(cb), C.byref(s2)))
        return dict(AHat=A, BHat=B, SigmaA=SA, SigmaB=SB, CA_diag=ca, CB_diag=cb, sigma2=s2.value)

    # ---- updates ----
    def step(self, which):
        self._chk(self._lib.vbmf_step(self._h, which))

    def note(self):
        """The library's note on the last call that returned OK ('' if none): vbmf_last_error after success carries e.g. the
        'eps below the resolution of d' remark of vbmf_run (include/vbmf_hip.h)."""
        m = self._lib.vbmf_last_error(self._h).decode()
        return m if m.startswith("note:") else ""

    def _warn_note(self):
        m = self.note()
        if m:
            import warnings
            warnings.warn(m, RuntimeWarning, stacklevel=3)

    def run(self, niter, eps=1e-6, est_covs=False, est_var=False, want_trace=False):
        it = C.c_int64(); d = C.c_double()
        tr = np.zeros((max(niter, 1), 4)) if want_trace else None
        self._chk(self._lib.vbmf_run(self._h, niter, eps, int(est_covs), int(est_var), C.byref(it), C.byref(d),
                                     _dptr(tr)))
        self._warn_note()
        return it.value, d.value, (tr[:it.value] if want_trace else None)

    def run_fixed_basis(self, niter):
        self._chk(self._lib.vbmf_run_fixed_basis(self._h, int(niter)))

    def sparse_run_fixed_basis(self, niter):
        self._chk(self._lib.vbmf_sparse_run_fixed_basis(self._h, int(niter)))

    def YHat(self):
        out = np.empty((self.L, self.M), order="F")
        self._chk(self._lib.vbmf_get_YHat(self._h, _dptr(out), self.L))
        return out

    def elbo(self):
        v = C.c_double()
        self._chk(self._lib.vbmf_elbo(self._h, C.byref(v)))
        return v.value

    # ---- ARD-sparse variant (variant=VBMF_VARIANT_SPARSE_DIAG) ----
    def sparse_set_state(self, ATVecHat, diagSigmaATVec, CA, beta, BHat, SigmaB, CB, delta, sigmaHat, zeta, hyper,
                         labels0=(), H1=0):
        n = self.M * self.H
        vecs = [np.ascontiguousarray(v, dtype=np.float64) for v in (ATVecHat, diagSigmaATVec, CA, beta)]
        for v in vecs:
            if v.shape != (n,):
                raise ValueError("vec(A')-shaped arguments must have length M*H")
        B = _fcol(BHat, (self.L, self.H)); SB = _fcol(SigmaB, (self.H, self.H))
        cb = np.ascontiguousarray(CB, dtype=np.float64); dl = np.ascontiguousarray(delta, dtype=np.float64)
        hp = VbmfSparseHyper(*[float(hyper[k]) for k in ("alpha0", "beta0", "gamma0", "delta0", "eta0", "zeta0")])
        lab = np.ascontiguousarray(labels0, dtype=np.int64)
        self._chk(self._lib.vbmf_sparse_set_state(self._h, _dptr(vecs[0]), _dptr(vecs[1]), _dptr(vecs[2]), _dptr(vecs[3]),
                                                  _dptr(B), self.L, _dptr(SB), _dptr(cb), _dptr(dl), float(sigmaHat),
                                                  float(zeta), C.byref(hp), lab.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  lab.size, int(H1)))

    def sparse_get_state(self, want_B=True):
        n = self.M * self.H
        a, ds, ca, be = (np.empty(n) for _ in range(4))
        sa = np.empty(self.H); B = np.empty((self.L, self.H), order="F") if want_B else None
        SB = np.empty((self.H, self.H), order="F"); cb = np.empty(self.H); dl = np.empty(self.H)
        sh, ze = C.c_double(), C.c_double()
        self._chk(self._lib.vbmf_sparse_get_state(self._h, _dptr(a), _dptr(ds), _dptr(ca), _dptr(be), _dptr(sa), _dptr(B),
                                                  self.L, _dptr(SB), _dptr(cb), _dptr(dl), C.byref(sh), C.byref(ze)))
        return dict(ATVecHat=a, diagSigmaATVec=ds, CA=ca, beta=be, SigmaA_diag=sa, BHat=B, SigmaB=SB, CB=cb, delta=dl,
                    sigmaHat=sh.value, zeta=ze.value)

    def sparse_step(self, which):
        self._chk(self._lib.vbmf_sparse_step(self._h, which))

    def sparse_run(self, niter, eps=1e-6, est_cb=True, want_trace=False):
        it = C.c_int64(); d = C.c_double()
        tr = np.zeros((max(niter, 1), 4)) if want_trace else None
        self._chk(self._lib.vbmf_sparse_run(self._h, niter, eps, int(est_cb), C.byref(it), C.byref(d), _dptr(tr)))
        self._warn_note()
        return it.value, d.value, (tr[:it.value] if want_trace else None)

    def sparse_set_noise_rows(self, sigmaVecHat, zetaVec, etaVec):
        s = np.ascontiguousarray(sigmaVecHat, dtype=np.float64); z = np.ascontiguousarray(zetaVec, dtype=np.float64)
        if s.shape != (self.L,) or z.shape != (self.L,):
            raise ValueError("sigmaVecHat and zetaVec must have length L")
        self._chk(self._lib.vbmf_sparse_set_noise_rows(self._h, _dptr(s), _dptr(z), float(etaVec)))

    def sparse_get_noise_rows(self):
        s, z = np.empty(self.L), np.empty(self.L)
        self._chk(self._lib.vbmf_sparse_get_noise_rows(self._h, _dptr(s), _dptr(z)))
        return s, z

    def sparse_set_full_cov(self, on=True):
        self._chk(self._lib.vbmf_sparse_set_full_cov(self._h, int(bool(on))))

    def sparse_set_SigmaA(self, SigmaA):
        S = _fcol(SigmaA, (self.H, self.H))
        self._chk(self._lib.vbmf_sparse_set_SigmaA(self._h, _dptr(S)))

    def sparse_get_SigmaA(self):
        S = np.empty((self.H, self.H), order="F")
        self._chk(self._lib.vbmf_sparse_get_SigmaA(self._h, _dptr(S)))
        return S

    def sparse_lower_bound(self, clamp=True):
        v = C.c_double()
        self._chk(self._lib.vbmf_sparse_lower_bound(self._h, int(clamp), C.byref(v)))
        return v.value

    def sparse_lower_bound_trimmed(self, trim=1e-1, clamp=True):
        v = C.c_double()
        self._chk(self._lib.vbmf_sparse_lower_bound_trimmed(self._h, int(clamp), float(trim), C.byref(v)))
        return v.value

    # ---- two-group ARD variant (variant=VBMF_VARIANT_DUAL_DIAG) ----
    def dual_set_priors(self, H0, alpha00, beta00, alpha01, beta01, alpha0=None, alpha1=None):
        alpha0 = alpha00 + 0.5 if alpha0 is None else alpha0
        alpha1 = alpha01 + 0.5 if alpha1 is None else alpha1
        self._chk(self._lib.vbmf_dual_set_priors(self._h, int(H0), float(alpha00), float(beta00), float(alpha01), float(beta01),
                                                 float(alpha0), float(alpha1)))

    def dual_get_priors(self):
        h0 = C.c_int64(); v = np.empty(6)
        self._chk(self._lib.vbmf_dual_get_priors(self._h, C.byref(h0), _dptr(v)))
        return h0.value, dict(alpha00=v[0], beta00=v[1], alpha01=v[2], beta01=v[3], alpha0=v[4], alpha1=v[5])

    def dual_run(self, niter, eps=1e-6, est_cb=True, est_priors=True, want_trace=False):
        it = C.c_int64(); d = C.c_double()
        tr = np.zeros((max(niter, 1), 4)) if want_trace else None
        self._chk(self._lib.vbmf_dual_run(self._h, niter, eps, int(est_cb), int(est_priors), C.byref(it), C.byref(d), _dptr(tr)))
        return it.value, d.value, (tr[:it.value] if want_trace else None)

    # ---- three-group ARD variant (variant=VBMF_VARIANT_TRIAL_DIAG) ----
    TRIAL_KEYS = ("alpha01", "beta01", "alpha02", "beta02", "alpha03", "beta03", "alpha1", "alpha2", "alpha3")

    def trial_set_priors(self, H0, M0, priors):
        v = np.array([float(priors[k]) for k in self.TRIAL_KEYS])
        self._chk(self._lib.vbmf_trial_set_priors(self._h, int(H0), int(M0), _dptr(v)))

    def trial_get_priors(self):
        h0, m0 = C.c_int64(), C.c_int64(); v = np.empty(9)
        self._chk(self._lib.vbmf_trial_get_priors(self._h, C.byref(h0), C.byref(m0), _dptr(v)))
        return h0.value, m0.value, {k: float(v[i]) for i, k in enumerate(self.TRIAL_KEYS)}

    def trial_run(self, niter, eps=1e-6, est_cb=True, est_priors=True, want_trace=False):
        it = C.c_int64(); d = C.c_double()
        tr = np.zeros((max(niter, 1), 4)) if want_trace else None
        self._chk(self._lib.vbmf_trial_run(self._h, niter, eps, int(est_cb), int(est_priors), C.byref(it), C.byref(d), _dptr(tr)))
        return it.value, d.value, (tr[:it.value] if want_trace else None)

    # ---- multi-GPU ----
    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = lib().vbmf_comm_unique_id(buf)
        if rc != 0:
            raise VbmfError(rc, "ncclGetUniqueId failed")
        return bytes(buf.raw)

    def comm_init(self, uid):
        buf = C.create_string_buffer(bytes(uid), UNIQUE_ID_BYTES)
        self._chk(self._lib.vbmf_comm_init(self._h, buf))

    def comm_set_transport(self, fn):
        """Bring-up transport instead of RCCL: fn(buf_ptr, count, is_double, stream_ptr) -> 0 sums a device buffer
        over the ranks in place (see include/vbmf_hip.h).  The ctypes thunk is kept alive with the context."""
        def thunk(user, buf, count, is_double, stream):
            try:
                return int(fn(buf, count, bool(is_double), stream) or 0)
            except Exception:            # never let a Python exception unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._ar_thunk = ALLREDUCE_FN(thunk)
        self._chk(self._lib.vbmf_comm_set_transport(self._h, self._ar_thunk, None))

    # ---- measurement ----
    def profile_enable(self, on=True):
        self._chk(self._lib.vbmf_profile_enable(self._h, int(on)))

    def profile_read(self, reset=True):
        out = np.zeros(8)
        self._chk(self._lib.vbmf_profile_read(self._h, _dptr(out), int(reset)))
        return dict(pass1_ms=out[0], pass1_n=int(out[1]), pass2_ms=out[2], pass2_n=int(out[3]))

    def pass_bytes(self, p):
        v = C.c_double()
        self._chk(self._lib.vbmf_pass_bytes(self._h, p, C.byref(v)))
        return v.value

    def peek(self, what, nwords, offset=0, dtype=np.uint32):
        out = np.zeros(nwords, dtype=np.uint32)
        self._chk(self._lib.vbmf_debug_peek(self._h, what, out.ctypes.data_as(C.POINTER(C.c_uint32)), nwords, offset))
        return out.view(dtype)

    def chain_us(self):
        """Last durations (microseconds) of the in-launch control chain's parts."""
        v = self.peek(PEEK_CHAIN, 16, dtype=np.uint64)
        return dict(zip(("ctrl_end", "SigmaA", "lambda_max_dB_and_loop", "SigmaB", "epilogue_table", "epilogue_tiles", "epilogue_fold"),
                        (float(x) * 0.01 for x in v)))

    def dims(self):
        v = self.peek(PEEK_DIMS, 16, dtype=np.int32)
        keys = ["Hp", "NH", "mode", "XT1", "KS1", "nsplit1", "sps1", "XT2", "KS2", "nsplit2", "sps2", "kstep", "npart", "narrow",
                "streamk_per", "streamk_grid"]       # segment-list plan of the Y*A pass: pieces per cut block (0: off), segments = workgroups
        return dict(zip(keys, (int(x) for x in v)))

    def time_pass(self, p, iters=10):
        v = C.c_double()
        self._chk(self._lib.vbmf_debug_time_pass(self._h, p, iters, C.byref(v)))
        return v.value

    def lambda_max(self, G):
        """lambda_max of the symmetric PSD H x H matrix G by the device kernel the run loop uses for this rank; (value, kernel us)."""
        G = np.asfortranarray(G, dtype=np.float64)
        assert G.shape == (self.H, self.H)
        v, us = C.c_double(), C.c_double()
        self._chk(self._lib.vbmf_debug_lambda_max(self._h, _dptr(G), C.byref(v), C.byref(us)))
        return v.value, us.value

    def sync(self):
        self._chk(self._lib.vbmf_device_sync(self._h))

    def debug_set(self, what, value):
        self._chk(self._lib.vbmf_debug_set(self._h, int(what), int(value)))
