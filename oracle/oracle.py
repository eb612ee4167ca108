"""ctypes binding of the CPU oracle (oracle/ssoracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (suitesparse_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libssoracle.so")
_lib = None

I64P = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
F64P = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ssoracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.orc_analyze.restype = C.c_void_p
    L.orc_analyze.argtypes = [C.c_int64, I64P, I64P, C.c_int, C.c_void_p, C.c_int,
                              C.c_void_p, C.c_void_p]
    L.orc_factorize.restype = C.c_int
    L.orc_factorize.argtypes = [C.c_void_p, I64P, I64P, F64P, C.c_int, C.c_double, C.c_int]
    L.orc_factorize_complex.restype = C.c_int
    L.orc_factorize_complex.argtypes = [C.c_void_p, I64P, I64P, F64P, C.c_void_p, C.c_int, C.c_double, C.c_int]
    L.orc_solve_complex.argtypes = [C.c_void_p, F64P, F64P, C.c_int64]
    L.orc_bind_blas_complex.restype = C.c_int
    L.orc_bind_blas_complex.argtypes = [C.c_char_p, C.c_char_p]
    L.orc_free.argtypes = [C.c_void_p]
    for name in ("orc_lsolve", "orc_ltsolve"):
        getattr(L, name).argtypes = [C.c_void_p, F64P, C.c_int64]
    L.orc_solve.argtypes = [C.c_void_p, F64P, F64P, C.c_int64]
    L.orc_sparent.argtypes = [C.c_void_p, I64P]
    L.orc_relmap_to_parent.argtypes = [C.c_void_p, I64P]
    L.orc_update_stats.argtypes = [C.c_void_p, F64P]
    L.orc_bind_blas.restype = C.c_int
    L.orc_bind_blas.argtypes = [C.c_char_p, C.c_char_p]
    for name in ("orc_n", "orc_nsuper", "orc_ssize", "orc_xsize", "orc_maxcsize",
                 "orc_maxesize", "orc_minor"):
        getattr(L, name).restype = C.c_int64
        getattr(L, name).argtypes = [C.c_void_p]
    L.orc_status.restype = C.c_int
    L.orc_status.argtypes = [C.c_void_p]
    for name in ("orc_fl", "orc_lnz", "orc_exec_flops"):
        getattr(L, name).restype = C.c_double
        getattr(L, name).argtypes = [C.c_void_p]
    for name in ("orc_perm", "orc_colcount", "orc_parent", "orc_super", "orc_pi", "orc_px",
                 "orc_s", "orc_x", "orc_calls"):
        getattr(L, name).restype = C.c_void_p
        getattr(L, name).argtypes = [C.c_void_p]
    _lib = L
    return L


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    ct = C.c_int64 if dtype == np.int64 else C.c_double
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))


class OracleFactor:
    """Owns an orc_factor; exposes the reference's cholmod_factor supernodal
    fields (CHOLMOD/Include/cholmod_core.h:1673-1798) as numpy views."""

    def __init__(self, n, Ap, Ai, stype=-1, perm=None, postorder=True,
                 nrelax=None, zrelax=None):
        L = lib()
        self.Ap = np.ascontiguousarray(Ap, dtype=np.int64)
        self.Ai = np.ascontiguousarray(Ai, dtype=np.int64)
        self.stype = stype
        pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        nr = None if nrelax is None else np.ascontiguousarray(nrelax, dtype=np.int64)
        zr = None if zrelax is None else np.ascontiguousarray(zrelax, dtype=np.float64)
        self._h = L.orc_analyze(n, self.Ap, self.Ai, stype,
                                None if pp is None else pp.ctypes.data, int(postorder),
                                None if nr is None else nr.ctypes.data,
                                None if zr is None else zr.ctypes.data)
        if not self._h:
            raise RuntimeError("orc_analyze failed")
        if L.orc_status(self._h) < 0:
            st = L.orc_status(self._h)
            L.orc_free(self._h)
            self._h = None
            raise RuntimeError(f"orc_analyze status {st}")
        self.n = n

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free(self._h)
            self._h = None

    # --- scalars
    nsuper = property(lambda s: lib().orc_nsuper(s._h))
    ssize = property(lambda s: lib().orc_ssize(s._h))
    xsize = property(lambda s: lib().orc_xsize(s._h))
    maxcsize = property(lambda s: lib().orc_maxcsize(s._h))
    maxesize = property(lambda s: lib().orc_maxesize(s._h))
    minor = property(lambda s: lib().orc_minor(s._h))
    status = property(lambda s: lib().orc_status(s._h))
    fl = property(lambda s: lib().orc_fl(s._h))
    lnz = property(lambda s: lib().orc_lnz(s._h))
    exec_flops = property(lambda s: lib().orc_exec_flops(s._h))
    # --- arrays (views into the C object)
    Perm = property(lambda s: _arr(lib().orc_perm(s._h), s.n, np.int64))
    ColCount = property(lambda s: _arr(lib().orc_colcount(s._h), s.n, np.int64))
    Parent = property(lambda s: _arr(lib().orc_parent(s._h), s.n, np.int64))
    super = property(lambda s: _arr(lib().orc_super(s._h), s.nsuper + 1, np.int64))
    pi = property(lambda s: _arr(lib().orc_pi(s._h), s.nsuper + 1, np.int64))
    px = property(lambda s: _arr(lib().orc_px(s._h), s.nsuper + 1, np.int64))
    s = property(lambda s: _arr(lib().orc_s(s._h), s.ssize, np.int64))
    x = property(lambda s: _arr(lib().orc_x(s._h), s.xsize, np.float64))
    calls = property(lambda s: _arr(lib().orc_calls(s._h), 4, np.int64))

    def factorize(self, Ax, beta=0.0, quick_return=False, Ap=None, Ai=None, stype=None):
        Ap = self.Ap if Ap is None else np.ascontiguousarray(Ap, dtype=np.int64)
        Ai = self.Ai if Ai is None else np.ascontiguousarray(Ai, dtype=np.int64)
        st = self.stype if stype is None else stype
        return lib().orc_factorize(self._h, Ap, Ai, np.ascontiguousarray(Ax, dtype=np.float64),
                                   st, float(beta), int(quick_return))

    def factorize_complex(self, Ax, beta=0.0, quick_return=False, zomplex=False):
        """Hermitian A with complex values (numpy complex128 array, one value per stored
        entry); zomplex=True hands the real and imaginary parts over as two arrays, as
        CHOLMOD_ZOMPLEX does.  L.x becomes complex (see xc)."""
        Ax = np.ascontiguousarray(Ax, dtype=np.complex128)
        if zomplex:
            re = np.ascontiguousarray(Ax.real)
            im = np.ascontiguousarray(Ax.imag)
            st = lib().orc_factorize_complex(self._h, self.Ap, self.Ai, re, im.ctypes.data,
                                             self.stype, float(beta), int(quick_return))
        else:
            st = lib().orc_factorize_complex(self._h, self.Ap, self.Ai, Ax.view(np.float64), None,
                                             self.stype, float(beta), int(quick_return))
        self._complex = True
        return st

    xc = property(lambda s: _arr(lib().orc_x(s._h), 2 * s.xsize, np.float64).view(np.complex128))

    def solve_complex(self, b):
        b = np.ascontiguousarray(b, dtype=np.complex128)
        nrhs = 1 if b.ndim == 1 else b.shape[0]
        x = np.empty_like(b)
        lib().orc_solve_complex(self._h, b.view(np.float64).reshape(-1), x.view(np.float64).reshape(-1), nrhs)
        return x

    def solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        nrhs = 1 if b.ndim == 1 else b.shape[0]
        x = np.empty_like(b)
        lib().orc_solve(self._h, b.reshape(-1), x.reshape(-1), nrhs)
        return x

    def lsolve(self, y):
        y = np.array(y, dtype=np.float64, order="C")
        lib().orc_lsolve(self._h, y.reshape(-1), 1 if y.ndim == 1 else y.shape[0])
        return y

    def ltsolve(self, y):
        y = np.array(y, dtype=np.float64, order="C")
        lib().orc_ltsolve(self._h, y.reshape(-1), 1 if y.ndim == 1 else y.shape[0])
        return y

    def sparent(self):
        out = np.empty(max(self.nsuper, 1), dtype=np.int64)
        lib().orc_sparent(self._h, out)
        return out[:self.nsuper]

    def relmap_to_parent(self):
        out = np.full(max(self.ssize - self.n, 1), -1, dtype=np.int64)
        lib().orc_relmap_to_parent(self._h, out)
        return out[:max(self.ssize - self.n, 0)]

    def update_stats(self):
        out = np.zeros(5)
        lib().orc_update_stats(self._h, out)
        return dict(updates=out[0], update_flops=out[1], scatter_elems=out[2],
                    panel_read_elems=out[3], panel_flops=out[4])

    def lower_mask(self):
        """Boolean mask over x selecting the lower-trapezoid entries (the strictly
        upper part of each diagonal block is dead space, SURVEY.md 8d parity)."""
        m = np.ones(self.xsize, dtype=bool)
        sup, pi, px = self.super, self.pi, self.px
        for s in range(self.nsuper):
            nscol = int(sup[s + 1] - sup[s])
            nsrow = int(pi[s + 1] - pi[s])
            if nscol > 1:
                blk = m[px[s]:px[s] + nsrow * nscol].reshape(nscol, nsrow)
                iu = np.triu_indices(nscol, k=1)
                # column j (first index), row i (second): upper means i < j
                blk[iu[1], iu[0]] = False
        return m


def bind_blas():
    """Try to bind an LP64 BLAS/LAPACK for the cpu_baseline leg.  Returns a
    description string or None (built-in C kernels are used then)."""
    import glob
    cands = []
    try:
        import scipy
        d = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
        cands += [(p, b"scipy_") for p in glob.glob(os.path.join(d, "libscipy_openblas-*.so"))]
    except Exception:
        pass
    cands += [(p, b"") for p in ("libopenblas.so.0", "libopenblas.so", "libmkl_rt.so",
                                 "/opt/conda/lib/libmkl_rt.so", "libblis.so")]
    for path, prefix in cands:
        try:
            if lib().orc_bind_blas(path.encode(), prefix):
                lib().orc_bind_blas_complex(path.encode(), prefix)
                return f"{os.path.basename(path)} (prefix '{prefix.decode()}')"
        except Exception:
            continue
    return None
