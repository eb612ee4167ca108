"""The reference's own test inputs through the supernodal path: every matrix of CHOLMOD/Tcov/Matrix and CHOLMOD/Demo/Matrix
(copied as data to tests/golden/tcov, tests/golden/demo) that can reach cholmod_l_super_numeric, driven the way
CHOLMOD/Tcov/cm.c:1203 (do_matrix) and Tcov/solve.c:35-172 (test_solver) drive it:

  * square symmetric / Hermitian files as they are (upper- or lower-stored, real, complex, zomplex for the z* files, pattern);
    rectangular and unsymmetric files as A*A' + beta*I with the demo's beta = 1e-6 (CHOLMOD/Demo/cholmod_l_demo.c:88, :279-285);
  * Common->supernodal = CHOLMOD_SUPERNODAL, natural ordering, crossed with postorder 0 / 1 and with the default relaxation
    against nrelax = zrelax = 0 ("no relaxed supernodes", Tcov/solve.c:105-121);
  * checked against the oracle (oracle/ssoracle.c) on the same input: Perm / ColCount / super / pi / px / s bit-exact,
    ||L - L_ref|| / ||L_ref|| < 1e-12 over the lower trapezoids, dead upper triangles exactly zero, status and L->minor
    equal -- on the matrices that are not positive definite (3singular, c3singular, z3singular, 1_0, the indefinite ones) the
    zeroed tail of L must be the oracle's, entry for entry (t_cholmod_super_numeric.c:883-968);
  * on the positive definite ones: ||A X - B||_1 / ||B||_1 for nrhs in {1, 2, 5, n} (Tcov/solve.c:174 sweeps nrhs; B = A*Z for
    a known Z as Tcov/cm.c:591-636), against a threshold scaled by the factor's own rcond, and X against the oracle's solve.

One function runs both legs: use_gpu = 0 is the product's CPU path (runs everywhere), use_gpu = 1 the HIP path (-m gpu)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from matrix_files import read_file, to_lower
from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL_L = 1e-12
BETA = 1e-6
EPS = np.finfo(float).eps


def _candidates():
    out = []
    for d in ("tcov", "demo"):
        for f in sorted(os.listdir(os.path.join(GOLD, d))):
            m = read_file(os.path.join(GOLD, d, f))
            if m["kind"] != "sparse" or max(m["nrow"], m["ncol"]) > 20000:
                continue
            out.append((d, f))
    return out


FILES = _candidates()
# the two copies of bcsstk01 / afiro etc. are kept: they are different files of the reference (lower / upper, .tri / .mtx)
VARIANTS = [("default", 0), ("default", 1), ("norelax", 0), ("norelax", 1)]


def _load(d, f):
    """-> dict: n, the matrix as handed to the library (Ap, Ai, Ax, stype, rect = scipy matrix or None), and the symmetric
    matrix the factorization is OF as lower CSC (Lp, Li, Lx) + beta, for the oracle and the residuals."""
    m = read_file(os.path.join(GOLD, d, f))
    zomplex = f.startswith("z")
    if m["stype"] != 0:
        n = m["nrow"]
        _, Lp, Li, Lx = to_lower(m)
        return dict(n=n, Ap=m["Ap"], Ai=m["Ai"], Ax=m["Ax"], stype=m["stype"], rect=None, Lp=Lp, Li=Li, Lx=Lx, beta=0.0,
                    cx=m["xtype"] == "complex", zomplex=zomplex)
    # unsymmetric / rectangular: A*A' + beta*I
    nrow, ncol = m["nrow"], m["ncol"]
    M = sp.csc_matrix((m["Ax"], m["Ai"], m["Ap"]), shape=(nrow, ncol))
    cx = m["xtype"] == "complex"
    Cm = sp.tril(sp.csc_matrix(M @ M.conj().T), format="csc")
    # (scipy keeps structural entries that cancel to zero: the symbolic pattern of A*A')
    Cm.sort_indices()
    return dict(n=nrow, Ap=m["Ap"], Ai=m["Ai"], Ax=m["Ax"], stype=0, rect=M, Lp=Cm.indptr.astype(np.int64),
                Li=Cm.indices.astype(np.int64), Lx=Cm.data.astype(np.complex128 if cx else np.float64), beta=BETA, cx=cx,
                zomplex=zomplex)


def _full(case):
    """The matrix being factorized, dense-free: scipy CSC with both triangles (Hermitian)."""
    n = case["n"]
    Lo = sp.csc_matrix((case["Lx"], case["Li"], case["Lp"]), shape=(n, n))
    D = sp.diags(Lo.diagonal().real if case["cx"] else Lo.diagonal())
    St = sp.tril(Lo, -1)
    return (St + St.conj().T + D + case["beta"] * sp.identity(n)).tocsc()


def _library_matrix(S, case):
    if case["rect"] is None:
        return S.sparse(case["n"], case["Ap"], case["Ai"], case["Ax"], case["stype"], zomplex=case["zomplex"])
    M = case["rect"]
    nrow, ncol = M.shape
    nz = int(case["Ap"][-1])
    xtype = (ch.ZOMPLEX if case["zomplex"] else ch.COMPLEX) if case["cx"] else ch.REAL
    A = S.L.cholmod_l_allocate_sparse(nrow, ncol, max(nz, 1), 1, 1, 0, xtype, C.byref(S.cm))
    assert A
    a = A.contents
    ch._view(a.p, ncol + 1, C.c_int64, np.int64)[:] = case["Ap"]
    if nz:
        ch._view(a.i, nz, C.c_int64, np.int64)[:] = case["Ai"]
        if not case["cx"]:
            ch._view(a.x, nz, C.c_double, np.float64)[:] = case["Ax"]
        elif case["zomplex"]:
            ch._view(a.x, nz, C.c_double, np.float64)[:] = case["Ax"].real
            ch._view(a.z, nz, C.c_double, np.float64)[:] = case["Ax"].imag
        else:
            ch._view(a.x, 2 * nz, C.c_double, np.float64)[:] = np.ascontiguousarray(case["Ax"], dtype=np.complex128).view(np.float64)
    return A


def run_case(d, f, relax, postorder, use_gpu, beta=None, quick=False):
    case = _load(d, f)
    if beta is not None:
        case["beta"] = beta
    n = case["n"]
    nrelax = zrelax = None
    S = ch.Session(use_gpu=use_gpu, postorder=bool(postorder))          # supernodal forced, natural ordering
    S.cm.error_handler = ch.ERRFUNC(0)
    S.cm.quick_return_if_not_posdef = int(quick)
    if relax == "norelax":
        nrelax, zrelax = [0, 0, 0], [0.0, 0.0, 0.0]
        for k in range(3):
            S.cm.nrelax[k] = 0
            S.cm.zrelax[k] = 0.0
    A = _library_matrix(S, case)
    if n == 0:
        # 0-by-0 (and 0-by-k): the analysis succeeds on an empty factor, nothing to compare (Tcov feeds these to every routine)
        Lf = S.L.cholmod_l_analyze(A, C.byref(S.cm))
        assert Lf and S.cm.status == ch.OK and Lf.contents.n == 0
        b2 = (C.c_double * 2)(case["beta"], 0.0)
        assert S.L.cholmod_l_factorize_p(A, C.byref(b2), None, 0, Lf, C.byref(S.cm)) == 1 and S.cm.status == ch.OK
        S.free_factor(Lf)
        S.free_sparse(A)
        assert S.cm.malloc_count == 0
        S.finish()
        return
    O = OracleFactor(n, case["Lp"], case["Li"], -1, perm=None, postorder=bool(postorder), nrelax=nrelax, zrelax=zrelax)
    ost = (O.factorize_complex(case["Lx"], beta=case["beta"], quick_return=quick) if case["cx"]
           else O.factorize(case["Lx"], beta=case["beta"], quick_return=quick))
    Lf = S.L.cholmod_l_analyze(A, C.byref(S.cm))
    assert Lf and S.cm.status == ch.OK, S.cm.status
    b2 = (C.c_double * 2)(case["beta"], 0.0)
    ok = S.L.cholmod_l_factorize_p(A, C.byref(b2), None, 0, Lf, C.byref(S.cm))
    assert ok == 1                                                       # TRUE also when not positive definite
    fv = ch.FactorView(Lf)
    for k in ("Perm", "ColCount", "super", "pi", "px", "s"):
        assert np.array_equal(getattr(fv, k), getattr(O, k)), k
    assert (fv.maxcsize, fv.maxesize) == (O.maxcsize, O.maxesize)
    assert fv.is_super and fv.is_ll and fv.xtype == (ch.COMPLEX if case["cx"] else ch.REAL)
    Ox = O.xc if case["cx"] else O.x
    mask = O.lower_mask()
    assert S.cm.status == (ch.NOT_POSDEF if ost == 1 else ch.OK), (S.cm.status, ost)
    assert fv.minor == O.minor
    assert np.all(fv.x[~mask] == 0)
    if ost == 1:
        # the zeroed tail is the oracle's, entry for entry; what was computed before the failing column agrees
        assert np.array_equal(fv.x[mask] == 0, Ox[mask] == 0)
        live = mask & (Ox != 0) & np.isfinite(Ox)
        if live.any():
            assert np.linalg.norm((fv.x - Ox)[live]) <= 1e-11 * np.linalg.norm(Ox[live])
        assert fv.minor < n
    else:
        didx = np.array([O.px[s] + j * (O.pi[s + 1] - O.pi[s]) + j for s in range(O.nsuper)
                         for j in range(O.super[s + 1] - O.super[s])], dtype=np.int64)
        dg = np.abs(Ox[didx])
        rcond = float((dg.min() / dg.max()) ** 2) if np.all(np.isfinite(dg)) and dg.max() > 0 else 0.0
        finite = np.all(np.isfinite(Ox[mask]))
        if finite:
            tol = max(TOL_L, 20 * EPS / max(rcond, 1e-300))
            assert np.linalg.norm((fv.x - Ox)[mask]) <= tol * np.linalg.norm(Ox[mask]), (rcond,)
            assert S.L.cholmod_l_check_factor(Lf, C.byref(S.cm)) == 1
            # cholmod_l_rcond: (min / max of L's own diagonal) ^ 2 (CHOLMOD/Cholesky/cholmod_rcond.c:120-150)
            dm = np.abs(fv.x[didx])
            assert abs(S.L.cholmod_l_rcond(Lf, C.byref(S.cm)) - (dm.min() / dm.max()) ** 2) <= 4 * EPS * rcond
        if finite and rcond > 1e-12:
            Af = _full(case)
            for nrhs in sorted({1, 2, 5, n} if n <= 120 else {1, 2, 5}):
                Z = (1.0 + np.arange(n)[None, :] / n + np.arange(nrhs)[:, None]).astype(np.complex128 if case["cx"] else np.float64)
                if case["cx"]:
                    Z = Z + 1j * (0.5 - np.arange(n)[None, :] / (2.0 * n))
                B = (Af @ Z.T).T
                X = S.solve(Lf, B if nrhs > 1 else B[0], zomplex=case["zomplex"] and case["cx"])
                X = X.reshape(nrhs, n)
                R = (Af @ X.T).T - B
                r1 = np.abs(R).sum(axis=1).max() / max(np.abs(B).sum(axis=1).max(), 1e-300)
                assert r1 <= max(1e-11, 50 * EPS / rcond), (nrhs, r1, rcond)
                Xo = (O.solve_complex(B) if case["cx"] else O.solve(B)).reshape(nrhs, n)
                assert np.linalg.norm(X - Xo) <= max(1e-10, 200 * EPS / rcond) * np.linalg.norm(Xo), nrhs
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


IDS = [f"{d}/{f}-{r}-post{p}" for d, f in FILES for r, p in VARIANTS]
PARAMS = [(d, f, r, p) for d, f in FILES for r, p in VARIANTS]


@pytest.mark.parametrize("d,f,relax,postorder", PARAMS, ids=IDS)
def test_reference_matrix_cpu_path(d, f, relax, postorder):
    run_case(d, f, relax, postorder, use_gpu=0)


@pytest.mark.gpu
@pytest.mark.parametrize("d,f,relax,postorder", PARAMS, ids=IDS)
def test_reference_matrix_hip_path(d, f, relax, postorder):
    assert ch.lib().cholmod_hip_probe() == 1, "no HIP device visible"
    run_case(d, f, relax, postorder, use_gpu=1)


# ---- the matrices that are not positive definite, once more: quick_return_if_not_posdef (the failing supernode is not
# refactorized up to the failing column, t_cholmod_super_numeric.c:905-925), and the rank-deficient rectangular files with
# beta = 0 where A*A' is EXACTLY singular (1_0 is a 1-by-0 matrix, a2 is 200-by-200 with no entry: a zero pivot in any
# arithmetic).  3_2 (3-by-2, rank 2) is not in the list: its third pivot is rounding noise around zero -- LAPACK's dpotrf
# gets +5.5e-17, the oracle's loop a non-positive value, the HIP kernels a positive one -- so whether it "is" positive
# definite is decided by the summation order, not by the algorithm
SINGULAR = [("tcov", f, None) for f in ("2lo.tri", "2up.tri", "3singular", "c3singular", "z3singular", "cha", "cha.mtx")] + \
           [("demo", "n5", None), ("tcov", "1_0", 0.0), ("tcov", "a2", 0.0)]
SING_PARAMS = [(d, f, b, q, p) for d, f, b in SINGULAR for q in (False, True) for p in (0, 1)]
SING_IDS = [f"{d}/{f}-beta{b}-quick{int(q)}-post{p}" for d, f, b, q, p in SING_PARAMS]


def _singular(d, f, beta, quick, postorder, use_gpu):
    case = _load(d, f)
    n = case["n"]
    O = OracleFactor(n, case["Lp"], case["Li"], -1, perm=None, postorder=bool(postorder))
    b = case["beta"] if beta is None else beta
    ost = O.factorize_complex(case["Lx"], beta=b, quick_return=quick) if case["cx"] else O.factorize(case["Lx"], beta=b, quick_return=quick)
    assert ost == 1 and O.minor < n, "the fixture is expected to be singular"
    run_case(d, f, "default", postorder, use_gpu, beta=beta, quick=quick)


@pytest.mark.parametrize("d,f,beta,quick,postorder", SING_PARAMS, ids=SING_IDS)
def test_not_positive_definite_reference_matrix_cpu_path(d, f, beta, quick, postorder):
    _singular(d, f, beta, quick, postorder, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("d,f,beta,quick,postorder", SING_PARAMS, ids=SING_IDS)
def test_not_positive_definite_reference_matrix_hip_path(d, f, beta, quick, postorder):
    assert ch.lib().cholmod_hip_probe() == 1, "no HIP device visible"
    _singular(d, f, beta, quick, postorder, 1)
