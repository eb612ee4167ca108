"""TEST INFRASTRUCTURE ONLY: the reference tree's own CSparse, compiled from its sources where they lie
(oracle/Makefile, target `ref` -> oracle/_ref/libcsparse_ref.so), called through ctypes.

CHOLMOD cannot be built in this image; CSparse can -- cs.h includes the C library only.  It holds the
reference's elimination tree (CSparse/Source/cs_etree.c), postorder (cs_post.c), column counts (cs_counts.c: the algorithm
CHOLMOD/Cholesky/cholmod_rowcolcounts.c implements too), symbolic analysis (cs_schol.c) and an up-looking numeric Cholesky
(cs_chol.c) -- another algorithm for the SAME factor L of P A P', so its output is reference-computed truth for

    the etree, the column counts, the pattern of L, the values of L

of the supernodal path.  `reference_cholesky` below runs exactly the steps of cs_schol with a permutation of the caller's
choice (cs_schol itself only knows natural / AMD): cs_pinv, cs_symperm, cs_etree, cs_post, cs_counts, then cs_chol."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libcsparse_ref.so")
csi = C.c_int64            # cs.h: csi = ptrdiff_t
_p = C.POINTER(csi)
_d = C.POINTER(C.c_double)


class CS(C.Structure):     # cs.h: struct cs_sparse
    _fields_ = [("nzmax", csi), ("m", csi), ("n", csi), ("p", _p), ("i", _p), ("x", _d), ("nz", csi)]


class CSS(C.Structure):    # cs.h: struct cs_symbolic
    _fields_ = [("pinv", _p), ("q", _p), ("parent", _p), ("cp", _p), ("leftmost", _p), ("m2", csi),
                ("lnz", C.c_double), ("unz", C.c_double)]


class CSN(C.Structure):    # cs.h: struct cs_numeric
    _fields_ = [("L", C.POINTER(CS)), ("U", C.POINTER(CS)), ("pinv", _p), ("B", _d)]


_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.cs_pinv.restype = _p
        L.cs_pinv.argtypes = [_p, csi]
        L.cs_symperm.restype = C.POINTER(CS)
        L.cs_symperm.argtypes = [C.POINTER(CS), _p, csi]
        L.cs_etree.restype = _p
        L.cs_etree.argtypes = [C.POINTER(CS), csi]
        L.cs_post.restype = _p
        L.cs_post.argtypes = [_p, csi]
        L.cs_counts.restype = _p
        L.cs_counts.argtypes = [C.POINTER(CS), _p, _p, csi]
        L.cs_chol.restype = C.POINTER(CSN)
        L.cs_chol.argtypes = [C.POINTER(CS), C.POINTER(CSS)]
        L.cs_spfree.restype = C.POINTER(CS)
        L.cs_spfree.argtypes = [C.POINTER(CS)]
        L.cs_nfree.restype = C.POINTER(CSN)
        L.cs_nfree.argtypes = [C.POINTER(CSN)]
        L.cs_free.restype = C.c_void_p
        L.cs_free.argtypes = [C.c_void_p]
        for f in ("cs_ipvec", "cs_pvec"):
            getattr(L, f).restype = csi
            getattr(L, f).argtypes = [_p, _d, _d, csi]
        for f in ("cs_lsolve", "cs_ltsolve"):
            getattr(L, f).restype = csi
            getattr(L, f).argtypes = [C.POINTER(CS), _d]
        _lib = L
    return _lib


def _upper(n, Ap, Ai, Ax, stype):
    """the upper triangle in CSC (what cs_schol / cs_chol read), from either stored triangle"""
    import scipy.sparse as sp
    A = sp.csc_matrix((np.asarray(Ax, dtype=np.float64), np.asarray(Ai), np.asarray(Ap)), shape=(n, n))
    if stype < 0:
        A = sp.csc_matrix(A.T)
    elif stype == 0:
        A = sp.triu(A, format="csc")
    A.sum_duplicates()
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)


def reference_cholesky(n, Ap, Ai, Ax, stype, perm=None, numeric=True, b=None):
    """-> dict(parent, colcount, lnz, Lp, Li, Lx, ok[, x]): etree and column counts of P A P' (perm: new -> old, as L->Perm),
    and -- numeric -- its Cholesky factor in CSC, all computed by the reference's CSparse.  ok = False: not positive
    definite (cs_chol returned NULL).  b (n or n x k, column-major columns): also x = A \\ b by the steps of cs_cholsol.c --
    cs_ipvec, cs_lsolve, cs_ltsolve, cs_pvec."""
    L = lib()
    Up, Ui, Ux = _upper(n, Ap, Ai, Ax, stype)
    A = CS(len(Ui), n, n, Up.ctypes.data_as(_p), Ui.ctypes.data_as(_p), Ux.ctypes.data_as(_d), -1)
    perm = np.arange(n, dtype=np.int64) if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
    out = {}
    pinv = L.cs_pinv(perm.ctypes.data_as(_p), n)               # cs_schol.c: S->pinv = cs_pinv (P, n)
    Cm = L.cs_symperm(C.byref(A), pinv, 0)                     #             C = cs_symperm (A, S->pinv, 0)
    parent = L.cs_etree(Cm, 0)                                 #             S->parent = cs_etree (C, 0)
    post = L.cs_post(parent, n)                                #             post = cs_post (S->parent, n)
    cnt = L.cs_counts(Cm, parent, post, 0)                     #             c = cs_counts (C, S->parent, post, 0)
    try:
        if not (pinv and Cm and parent and post and cnt):
            raise MemoryError("CSparse analysis")
        out["parent"] = np.ctypeslib.as_array(parent, shape=(n,)).copy() if n else np.zeros(0, dtype=np.int64)
        out["colcount"] = np.ctypeslib.as_array(cnt, shape=(n,)).copy() if n else np.zeros(0, dtype=np.int64)
        out["lnz"] = int(out["colcount"].sum())
        if numeric:
            cp = np.zeros(n + 1, dtype=np.int64)               #             S->lnz = cs_cumsum (S->cp, c, n)
            np.cumsum(out["colcount"], out=cp[1:])
            S = CSS(pinv, None, parent, cp.ctypes.data_as(_p), None, 0, float(cp[n]), float(cp[n]))
            N = L.cs_chol(C.byref(A), C.byref(S))
            out["ok"] = bool(N)
            if N:
                Lm = N.contents.L.contents
                nz = int(Lm.p[n])
                out["Lp"] = np.ctypeslib.as_array(Lm.p, shape=(n + 1,)).copy()
                out["Li"] = np.ctypeslib.as_array(Lm.i, shape=(max(nz, 1),))[:nz].copy()
                out["Lx"] = np.ctypeslib.as_array(Lm.x, shape=(max(nz, 1),))[:nz].copy()
                if b is not None:
                    B = np.atleast_2d(np.asarray(b, dtype=np.float64).T).copy()        # rows = right-hand sides
                    X = np.empty_like(B)
                    w = np.empty(n)
                    for r in range(B.shape[0]):
                        L.cs_ipvec(pinv, B[r].ctypes.data_as(_d), w.ctypes.data_as(_d), n)      # x = P*b
                        L.cs_lsolve(N.contents.L, w.ctypes.data_as(_d))                         # x = L\\x
                        L.cs_ltsolve(N.contents.L, w.ctypes.data_as(_d))                        # x = L'\\x
                        L.cs_pvec(pinv, w.ctypes.data_as(_d), X[r].ctypes.data_as(_d), n)       # b = P'*x
                    out["x"] = X[0] if np.ndim(b) == 1 else X.T.copy()
                L.cs_nfree(N)
    finally:
        for q in (pinv, parent, post, cnt):
            if q:
                L.cs_free(C.cast(q, C.c_void_p))
        if Cm:
            L.cs_spfree(Cm)
    return out
