"""CPU tests of the harness-level host API (reader, transposes, sdmult, norms),
mirroring what the reference's demo uses around the hot path
(CHOLMOD/Demo/cholmod_l_demo.c; format notes CHOLMOD/Check/cholmod_read.c:14-110)."""
import ctypes as C
import os

import numpy as np
import pytest

from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]


def _read(S, path):
    fp = libc.fopen(str(path).encode(), b"r")
    assert fp
    A = S.L.cholmod_l_read_sparse(fp, C.byref(S.cm))
    libc.fclose(fp)
    return A


def _dense_of(A):
    a = A.contents
    n, m = a.nrow, a.ncol
    p = ch._view(a.p, m + 1, C.c_int64, np.int64)
    nz = int(p[-1])
    i = ch._view(a.i, nz, C.c_int64, np.int64)
    x = ch._view(a.x, nz, C.c_double, np.float64)
    D = np.zeros((n, m))
    for j in range(m):
        for q in range(p[j], p[j + 1]):
            D[i[q], j] += x[q]
    if a.stype > 0:
        D = D + np.triu(D, 1).T
    elif a.stype < 0:
        D = D + np.tril(D, -1).T
    return D


def test_read_matrix_market_symmetric_and_triplet_forms(tmp_path):
    S = ch.Session(use_gpu=0)
    mm = tmp_path / "a.mtx"
    mm.write_text("%%MatrixMarket matrix coordinate real symmetric\n% comment\n3 3 4\n1 1 4.0\n2 1 -1.0\n"
                  "2 2 5.0\n3 3 6.0\n")
    A = _read(S, mm)
    assert A and A.contents.stype == 1                      # prefer_upper: returned upper
    ref = np.array([[4, -1, 0], [-1, 5, 0], [0, 0, 6.0]])
    assert np.array_equal(_dense_of(A), ref)
    S.free_sparse(A)
    # zero-based triplet with explicit stype and duplicates (summed)
    tri = tmp_path / "b.tri"
    tri.write_text("3 3 5 -1\n0 0 2.0\n0 0 2.0\n1 0 -1.0\n1 1 5.0\n2 2 6.0\n")
    S.cm.prefer_upper = 0
    B = _read(S, tri)
    assert B and B.contents.stype == -1
    assert np.array_equal(_dense_of(B), ref)
    assert S.L.cholmod_l_check_sparse(B, C.byref(S.cm)) == 1
    S.free_sparse(B)
    # unsymmetric file: stype inferred as 0
    gen = tmp_path / "c.tri"
    gen.write_text("2 3 3\n1 1 1.0\n2 3 2.0\n1 2 3.0\n")
    Cm = _read(S, gen)
    assert Cm and Cm.contents.stype == 0 and Cm.contents.nrow == 2 and Cm.contents.ncol == 3
    S.free_sparse(Cm)
    assert S.cm.malloc_count == 0
    S.finish()


def test_ptranspose_sdmult_and_norms_against_numpy():
    n, Ap, Ai, Ax = G.poisson3d(5)
    rng = np.random.default_rng(0)
    Ax = Ax + rng.standard_normal(Ax.size) * 0.01
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Ad = _dense_of(A)
    perm = rng.permutation(n).astype(np.int64)
    T = S.L.cholmod_l_ptranspose(A, 2, perm.ctypes.data, None, 0, C.byref(S.cm))
    assert T.contents.stype == 1 and T.contents.sorted == 1
    assert np.allclose(_dense_of(T), Ad[np.ix_(perm, perm)])
    X = rng.standard_normal((2, n))
    Xd = S.dense(X)
    Yd = S.dense(np.ones((2, n)))
    one = (C.c_double * 2)(2.0, 0.0)
    beta = (C.c_double * 2)(-1.0, 0.0)
    assert S.L.cholmod_l_sdmult(A, 0, C.byref(one), C.byref(beta), Xd, Yd, C.byref(S.cm)) == 1
    Y = S.dense_to_numpy(Yd)
    assert np.allclose(Y, 2.0 * (Ad @ X.T).T - 1.0)
    assert np.isclose(S.L.cholmod_l_norm_sparse(A, 0, C.byref(S.cm)), np.abs(Ad).sum(axis=1).max())
    assert np.isclose(S.L.cholmod_l_norm_sparse(A, 1, C.byref(S.cm)), np.abs(Ad).sum(axis=0).max())
    x1 = S.dense(X[0])
    assert np.isclose(S.L.cholmod_l_norm_dense(x1, 2, C.byref(S.cm)), np.linalg.norm(X[0]))
    assert np.isclose(S.L.cholmod_l_norm_dense(x1, 0, C.byref(S.cm)), np.abs(X[0]).max())
    assert np.isclose(S.L.cholmod_l_norm_dense(Xd, 1, C.byref(S.cm)), np.abs(X).sum(axis=1).max())
    for d in (Xd, Yd, x1):
        S.free_dense(d)
    S.free_sparse(T)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0 and S.cm.memory_inuse == 0
    S.finish()


def test_etree_postorder_rowcolcounts_entry_points():
    """The building blocks the reference's Tcov raw_factor test calls directly
    (CHOLMOD/Tcov/raw_factor.c:165-298)."""
    from oracle.oracle import OracleFactor
    n, Ap, Ai, Ax = G.poisson2d(7)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    U = S.L.cholmod_l_ptranspose(A, 0, None, None, 0, C.byref(S.cm))       # upper pattern
    parent = np.empty(n, dtype=np.int64)
    assert S.L.cholmod_l_etree(U, parent.ctypes.data, C.byref(S.cm)) == 1
    post = np.empty(n, dtype=np.int64)
    assert S.L.cholmod_l_postorder(parent.ctypes.data, n, None, post.ctypes.data, C.byref(S.cm)) == n
    cc = np.empty(n, dtype=np.int64)
    rc = np.empty(n, dtype=np.int64)
    first = np.empty(n, dtype=np.int64)
    level = np.empty(n, dtype=np.int64)
    assert S.L.cholmod_l_rowcolcounts(A, None, 0, parent.ctypes.data, post.ctypes.data, rc.ctypes.data,
                                      cc.ctypes.data, first.ctypes.data, level.ctypes.data,
                                      C.byref(S.cm)) == 1
    O = OracleFactor(n, Ap, Ai, -1, perm=None, postorder=False)
    assert np.array_equal(parent, O.Parent) and np.array_equal(cc, O.ColCount)
    assert rc.sum() == cc.sum() and S.cm.lnz == cc.sum() and S.cm.fl == (cc.astype(float) ** 2).sum()
    assert sorted(post.tolist()) == list(range(n))
    # etree on a lower-stored matrix is rejected as in the reference (cholmod_etree.c:205-210)
    assert S.L.cholmod_l_etree(A, parent.ctypes.data, C.byref(S.cm)) == 0 and S.cm.status == ch.INVALID
    S.free_sparse(U)
    S.free_sparse(A)
    S.finish()


@pytest.mark.parametrize("zomplex", [False, True])
def test_norm_dense_complex_and_zomplex(zomplex):
    """cholmod_l_norm_dense on the complex layouts cholmod_l_solve now returns (round-2 advice):
    |re + i im| per entry as CHOLMOD/MatrixOps/cholmod_norm.c:34-60, 2-norm of a column vector
    as :171-204 -- the reference demo's residual check on the complex path."""
    rng = np.random.default_rng(5)
    S = ch.Session(use_gpu=0)
    X = rng.standard_normal((3, 17)) + 1j * rng.standard_normal((3, 17))     # 3 columns of length 17
    Xd = S.dense(X, zomplex=zomplex)
    x1 = S.dense(X[0], zomplex=zomplex)
    assert np.isclose(S.L.cholmod_l_norm_dense(Xd, 0, C.byref(S.cm)), np.abs(X).sum(axis=0).max(), rtol=1e-14)
    assert np.isclose(S.L.cholmod_l_norm_dense(Xd, 1, C.byref(S.cm)), np.abs(X).sum(axis=1).max(), rtol=1e-14)
    assert np.isclose(S.L.cholmod_l_norm_dense(x1, 2, C.byref(S.cm)), np.linalg.norm(X[0]), rtol=1e-14)
    assert S.cm.status == ch.OK
    S.free_dense(Xd)
    S.free_dense(x1)
    S.finish()


def test_factorize_rejects_a_factor_analysed_for_spqr():
    """CHOLMOD_ANALYZE_FOR_SPQR leaves px[0] = 123456 and xsize = 1 (cholmod_super_symbolic.c:
    662-663, :749-771): cholmod_l_factorize on such a factor must return CHOLMOD_INVALID, not
    write at Lx + 123456 (round-2 advice: reproduced heap overflow on the CPU path)."""
    n, Ap, Ai, Ax = G.poisson3d(6)
    perm = np.ascontiguousarray(G.geometric_nd(6, 6, 6, 3))
    for use_gpu in (0, 1):
        S = ch.Session(use_gpu=use_gpu)
        S.cm.error_handler = ch.ERRFUNC(0)
        A = S.sparse(n, Ap, Ai, Ax, -1)
        Lf = S.L.cholmod_l_analyze_p2(0, A, perm.ctypes.data, None, 0, C.byref(S.cm))
        assert Lf and S.cm.status == ch.OK
        assert S.L.cholmod_l_factorize(A, Lf, C.byref(S.cm)) == 0
        assert S.cm.status == ch.INVALID
        assert ch.FactorView(Lf).xsize == 1 and not Lf.contents.x
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()


def test_check_sparse_refuses_what_the_reference_refuses():
    """cholmod_l_check_sparse, condition by condition (reference Check/cholmod_check.c:652-960): a valid matrix passes; every
    single defect makes it return FALSE with CHOLMOD_INVALID and leaves the matrix alone."""
    S = ch.Session(use_gpu=0)
    Ap = np.array([0, 2, 3, 5], dtype=np.int64)
    Ai = np.array([0, 2, 1, 0, 2], dtype=np.int64)
    Ax = np.arange(1.0, 6.0)

    def fresh(stype=0):
        return S.sparse(3, Ap, Ai, Ax, stype)

    def refused(A):
        ok = S.L.cholmod_l_check_sparse(A, C.byref(S.cm))
        st = S.cm.status
        return ok == 0 and st == ch.INVALID

    A = fresh()
    assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 1 and S.cm.status == ch.OK
    a = A.contents
    p = ch._view(a.p, 4, C.c_int64, np.int64)
    i = ch._view(a.i, 5, C.c_int64, np.int64)
    p[0] = 1 ; assert refused(A) ; p[0] = 0                         # p [0] must be zero
    p[3] = 7 ; assert refused(A) ; p[3] = 5                         # p [ncol] beyond nzmax
    p[1] = 4 ; p[2] = 3 ; assert refused(A) ; p[1] = 2               # a column that ends before it starts
    i[1] = 3 ; assert refused(A) ; i[1] = 2                         # row index out of range
    i[1] = -1 ; assert refused(A) ; i[1] = 2
    i[0], i[1] = 2, 0 ; assert refused(A)                           # sorted flag set, indices out of order
    a.sorted = 0
    assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 1       # ... fine for an unsorted matrix
    i[0], i[1] = 2, 2 ; assert refused(A)                           # duplicate row index in an unsorted column
    a.sorted = 1 ; assert refused(A)                                # (and "out of order" for a sorted one)
    i[0], i[1] = 0, 2
    assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 1
    a.itype = 0 ; assert refused(A) ; a.itype = 2                   # CHOLMOD_INT matrix handed to the _l_ routine
    a.dtype = 1 ; assert refused(A) ; a.dtype = 0                   # single precision
    a.xtype = 7 ; assert refused(A) ; a.xtype = ch.REAL
    keep = a.x ; a.x = None ; assert refused(A)                     # values missing for a real matrix
    a.xtype = 0 ; assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 1      # ... a pattern has none
    a.xtype = ch.REAL ; a.x = keep
    a.stype = 1 ; a.ncol = 2 ; assert refused(A) ; a.ncol = 3 ; a.stype = 0     # symmetric but not square
    # unpacked: counts per column, a negative count is an empty column, a count beyond nrow is refused
    nz = np.array([2, 1, 2], dtype=np.int64)
    a.packed = 0 ; a.nz = nz.ctypes.data
    assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 1
    nz[1] = -3 ; assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 1
    nz[2] = 4 ; assert refused(A)
    a.nz = None ; assert refused(A)                                 # nz array not present
    a.packed = 1
    assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 1
    assert S.L.cholmod_l_check_sparse(None, C.byref(S.cm)) == 0 and S.cm.status == ch.INVALID
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()
