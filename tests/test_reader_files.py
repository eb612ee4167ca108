"""The product's reader (csrc/host/io.c: cholmod_l_read_sparse / _read_dense / _read_matrix) on every file of the reference's
own test and demo directories (tests/golden/tcov = CHOLMOD/Tcov/Matrix, tests/golden/demo = CHOLMOD/Demo/Matrix), against the
test-side reader written from the format notes (tests/matrix_files.py; reference CHOLMOD/Check/cholmod_read.c:9-140): same
shape, stype (upper after prefer_upper), xtype, pattern and values; files the notes rule out come back NULL with
CHOLMOD_INVALID and leak nothing.  No GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from matrix_files import read_file
from suitesparse_amd import cholmod as ch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]

FILES = [(d, f) for d in ("tcov", "demo") for f in sorted(os.listdir(os.path.join(ROOT, "tests", "golden", d)))]
# dimensions only a 64-bit index holds and no machine here allocates: the reader must refuse or return them empty, not crash
HUGE = {"mega.tri", "zero", "tribig", "a1"}


def _csc(A):
    a = A.contents
    n, m = a.ncol, a.nrow
    Ap = ch._view(a.p, n + 1, C.c_int64, np.int64).copy()
    nz = int(Ap[n])
    Ai = ch._view(a.i, max(nz, 1), C.c_int64, np.int64)[:nz].copy()
    if a.xtype == ch.COMPLEX:
        Ax = ch._view(a.x, 2 * max(nz, 1), C.c_double, np.float64)[:2 * nz].copy().view(np.complex128)
    else:
        Ax = ch._view(a.x, max(nz, 1), C.c_double, np.float64)[:nz].copy()
    assert np.all(np.diff(Ap) >= 0)
    jj = np.repeat(np.arange(n), np.diff(Ap))
    key = jj * max(m, 1) + Ai
    assert np.all(np.diff(key) > 0)                                  # columns sorted, duplicates summed
    return m, n, Ap, Ai, Ax, jj


def _sparse_to_dense(A):
    m, n, Ap, Ai, Ax, jj = _csc(A)
    M = np.zeros((m, n), dtype=Ax.dtype)
    M[Ai, jj] = Ax
    return M


@pytest.mark.parametrize("d,f", FILES, ids=[f"{d}/{f}" for d, f in FILES])
def test_reader_matches_the_format_notes(d, f):
    path = os.path.join(ROOT, "tests", "golden", d, f)
    ref = read_file(path)
    S = ch.Session(use_gpu=0)
    S.cm.error_handler = ch.ERRFUNC(0)
    fp = libc.fopen(path.encode(), b"r")
    assert fp
    mtype = C.c_int(0)
    if f in HUGE:
        # as triplets (prefer == 0): nothing of size nrow / ncol is allocated
        got = S.L.cholmod_l_read_matrix(fp, 0, C.byref(mtype), C.byref(S.cm))
        libc.fclose(fp)
        assert got and mtype.value == 4 and S.cm.status == ch.OK
        T = C.cast(got, C.POINTER(ch.Triplet))
        assert (T.contents.nrow, T.contents.ncol, T.contents.nnz) == (ref["nrow"], ref["ncol"], int(ref["Ap"][-1]))
        tp = C.c_void_p(got)
        S.L.cholmod_l_free_triplet(C.byref(tp), C.byref(S.cm))
        assert S.cm.malloc_count == 0
        S.finish()
        return
    got = S.L.cholmod_l_read_matrix(fp, 2, C.byref(mtype), C.byref(S.cm))
    libc.fclose(fp)
    if ref["kind"] == "invalid":
        assert not got and S.cm.status == ch.INVALID, (f, ref["why"])
    elif ref["kind"] == "dense":
        assert got and mtype.value == 3
        X = C.cast(got, C.POINTER(ch.Dense))
        x = X.contents
        assert (x.nrow, x.ncol) == (ref["nrow"], ref["ncol"])
        if x.nrow * x.ncol:
            assert (x.xtype == ch.COMPLEX) == np.iscomplexobj(ref["X"])
            out = S.dense_to_numpy(X)
            out = out.reshape(x.ncol, x.nrow).T if x.ncol > 1 else out.reshape(-1, 1)
            assert np.array_equal(out, ref["X"])
        S.free_dense(X)
    else:
        assert got and mtype.value == 1, (S.cm.status, ref)
        A = C.cast(got, C.POINTER(ch.Sparse))
        a = A.contents
        assert (a.nrow, a.ncol) == (ref["nrow"], ref["ncol"])
        assert a.stype == (1 if ref["stype"] != 0 else 0)             # prefer == 2: symmetric files come back upper
        assert a.sorted and a.packed
        if ref["Ap"][-1]:
            assert (a.xtype == ch.COMPLEX) == (ref["xtype"] == "complex")
            m, n, Ap, Ai, Ax, jj = _csc(A)
            rj = np.repeat(np.arange(ref["ncol"]), np.diff(ref["Ap"]))
            ri, rx = ref["Ai"], ref["Ax"]
            if ref["stype"] < 0:
                ri, rj, rx = rj, ri, np.conj(rx)                       # lower file -> upper storage (A = A')
            o = np.lexsort((ri, rj))
            assert np.array_equal(Ai, ri[o]) and np.array_equal(jj, rj[o]) and np.array_equal(Ax, rx[o]), f
        S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


def test_read_sparse_and_read_dense_refuse_the_other_kind():
    S = ch.Session(use_gpu=0)
    S.cm.error_handler = ch.ERRFUNC(0)
    g = os.path.join(ROOT, "tests", "golden", "tcov")
    fp = libc.fopen(os.path.join(g, "fullrsa.mtx").encode(), b"r")
    assert not S.L.cholmod_l_read_sparse(fp, C.byref(S.cm)) and S.cm.status == ch.INVALID
    libc.fclose(fp)
    fp = libc.fopen(os.path.join(g, "r5lo").encode(), b"r")
    assert not S.L.cholmod_l_read_dense(fp, C.byref(S.cm)) and S.cm.status == ch.INVALID
    libc.fclose(fp)
    # prefer == 1: both triangles, stype 0
    fp = libc.fopen(os.path.join(g, "r5lo").encode(), b"r")
    mt = C.c_int(0)
    A = C.cast(S.L.cholmod_l_read_matrix(fp, 1, C.byref(mt), C.byref(S.cm)), C.POINTER(ch.Sparse))
    libc.fclose(fp)
    assert A and mt.value == 1 and A.contents.stype == 0
    M = _sparse_to_dense(A)
    assert np.array_equal(M, M.T) and np.count_nonzero(np.triu(M, 1)) > 0
    S.free_sparse(A)
    # prefer_binary: a symmetric pattern file keeps ones
    S.cm.prefer_binary = 1
    fp = libc.fopen(os.path.join(ROOT, "tests", "golden", "demo", "can___24.mtx").encode(), b"r")
    A = S.L.cholmod_l_read_sparse(fp, C.byref(S.cm))
    libc.fclose(fp)
    assert A and np.all(_sparse_to_dense(A)[np.nonzero(_sparse_to_dense(A))] == 1.0)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()
