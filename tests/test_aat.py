"""Unsymmetric input (A->stype == 0): cholmod_l_analyze / cholmod_l_factorize_p factorize A*A' + beta*I, as the reference does
(CHOLMOD/Cholesky/cholmod_factorize.c:197-224, CHOLMOD/Supernodal/t_cholmod_super_numeric.c:223-237, :385-418).  This build forms
tril (A*A') on the host (csrc/host/core.c: ssamd_aat) and takes the symmetric path; the factor is the same.  Checked against numpy on
the dense product: L*L' = P (A*A' + beta*I) P', the solve, a second factorization with new values, the expert entry point
cholmod_l_super_numeric (S, F, beta, L), and a column subset fset (A(:,f)*A(:,f)').  CPU path here, GPU path marked gpu."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from suitesparse_amd import cholmod as ch


def _rect(S, M, stype=0):
    """scipy CSC (m x n) -> library-owned cholmod_sparse, unsymmetric"""
    M = sp.csc_matrix(M)
    M.sort_indices()
    m, n = M.shape
    nz = M.nnz
    A = S.L.cholmod_l_allocate_sparse(m, n, max(nz, 1), 1, 1, stype, ch.REAL, C.byref(S.cm))
    assert A
    a = A.contents
    ch._view(a.p, n + 1, C.c_int64, np.int64)[:] = M.indptr
    ch._view(a.i, nz, C.c_int64, np.int64)[:] = M.indices
    ch._view(a.x, nz, C.c_double, np.float64)[:] = M.data
    return A


def _dense_L(fv):
    n = fv.n
    Ld = np.zeros((n, n))
    for s in range(fv.nsuper):
        k1, k2 = int(fv.super[s]), int(fv.super[s + 1])
        rows = fv.s[fv.pi[s]:fv.pi[s + 1]]
        blk = fv.x[fv.px[s]:fv.px[s] + len(rows) * (k2 - k1)].reshape(k2 - k1, len(rows)).T
        for j in range(k2 - k1):
            Ld[rows[j:], k1 + j] = blk[j:, j]
    return Ld


def _run(use_gpu):
    rng = np.random.default_rng(11)
    m, n = 60, 85
    M = sp.random(m, n, density=0.06, random_state=5, format="csc")
    M.data = rng.standard_normal(M.nnz)
    M = M + sp.csc_matrix((np.ones(m), (np.arange(m), np.arange(m))), shape=(m, n))      # (no empty row)
    beta = 0.25
    S = ch.Session(use_gpu=use_gpu, ordering="default")
    A = _rect(S, M)
    Lf = S.L.cholmod_l_analyze(A, C.byref(S.cm))
    assert Lf and S.cm.status == ch.OK and Lf.contents.n == m
    b2 = (C.c_double * 2)(beta, 0.0)
    assert S.L.cholmod_l_factorize_p(A, C.byref(b2), None, 0, Lf, C.byref(S.cm)) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    Cd = (M @ M.T).toarray() + beta * np.eye(m)
    P = fv.Perm
    Ld = _dense_L(fv)
    assert np.linalg.norm(Ld @ Ld.T - Cd[np.ix_(P, P)]) <= 1e-13 * np.linalg.norm(Cd)
    b = 1.0 + np.arange(m) / m
    x = S.solve(Lf, b)
    assert np.linalg.norm(Cd @ x - b) <= 1e-12 * np.linalg.norm(b)
    # new values, same pattern
    M2 = M.copy()
    M2.data = rng.standard_normal(M2.nnz) + 0.1
    A2 = _rect(S, M2)
    assert S.L.cholmod_l_factorize_p(A2, C.byref(b2), None, 0, Lf, C.byref(S.cm)) == 1 and S.cm.status == ch.OK
    Cd2 = (M2 @ M2.T).toarray() + beta * np.eye(m)
    Ld2 = _dense_L(ch.FactorView(Lf))
    assert np.linalg.norm(Ld2 @ Ld2.T - Cd2[np.ix_(P, P)]) <= 1e-13 * np.linalg.norm(Cd2)
    # the expert routine: S = A (p, :), F = S', L L' = S F + beta I (cholmod_super_numeric.c:97-308 with stype == 0)
    Sp = _rect(S, sp.csc_matrix(M2.toarray()[P, :]))
    Fp = S.L.cholmod_l_transpose(Sp, 1, C.byref(S.cm))
    assert Fp
    assert S.L.cholmod_l_super_numeric(Sp, Fp, C.byref(b2), Lf, C.byref(S.cm)) == 1 and S.cm.status == ch.OK
    Ld3 = _dense_L(ch.FactorView(Lf))
    assert np.linalg.norm(Ld3 - Ld2) <= 1e-13 * np.linalg.norm(Ld2)
    S.cm.error_handler = ch.ERRFUNC(0)
    assert S.L.cholmod_l_super_numeric(Sp, None, C.byref(b2), Lf, C.byref(S.cm)) == 0 and S.cm.status == ch.INVALID      # F is required
    # a column subset f: A(:,f)*A(:,f)' + beta*I (cholmod_analyze.c:402-418, cholmod_factorize.c:197-224), in the order given
    fset = np.ascontiguousarray(rng.permutation(n)[:50], dtype=np.int64)
    Lf2 = S.L.cholmod_l_analyze_p(A, None, fset.ctypes.data, len(fset), C.byref(S.cm))
    assert Lf2 and S.cm.status == ch.OK
    assert S.L.cholmod_l_factorize_p(A, C.byref(b2), fset.ctypes.data, len(fset), Lf2, C.byref(S.cm)) == 1 and S.cm.status == ch.OK
    fv2 = ch.FactorView(Lf2)
    Mf = M[:, fset]
    Cf = (Mf @ Mf.T).toarray() + beta * np.eye(m)
    Lf2d = _dense_L(fv2)
    P2 = fv2.Perm
    assert np.linalg.norm(Lf2d @ Lf2d.T - Cf[np.ix_(P2, P2)]) <= 1e-13 * np.linalg.norm(Cf)
    # an index listed twice, or outside the columns of A: CHOLMOD_INVALID
    bad = np.array([3, 7, 3], dtype=np.int64)
    assert S.L.cholmod_l_factorize_p(A, C.byref(b2), bad.ctypes.data, 3, Lf2, C.byref(S.cm)) == 0 and S.cm.status == ch.INVALID
    bad = np.array([3, n], dtype=np.int64)
    assert not S.L.cholmod_l_analyze_p(A, None, bad.ctypes.data, 2, C.byref(S.cm)) and S.cm.status == ch.INVALID
    S.free_factor(Lf2)
    for X in (Sp, Fp, A, A2):
        S.free_sparse(X)
    S.free_factor(Lf)
    assert S.cm.malloc_count == 0
    S.finish()


def test_unsymmetric_input_factorizes_aat_cpu_path():
    _run(0)


@pytest.mark.gpu
def test_unsymmetric_input_factorizes_aat_gpu_path():
    _run(1)
