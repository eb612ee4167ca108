"""Host-allocation fault loop, shaped after the reference's CHOLMOD/Tcov/memory.c:126-190: an allocator that fails its k-th
call is installed in SuiteSparse_config (include/SuiteSparse_config.h; every allocation of the host C layer goes through
it), and analyze -> factorize -> solve is repeated for k = 0, 1, 2, ... until a run gets through without a failure.  After
EVERY run Common->malloc_count and Common->memory_inuse must be back where they were, a failed call must have reported
CHOLMOD_OUT_OF_MEMORY, and a factorization that fails on a symbolic L must hand L back symbolic
(CHOLMOD/Supernodal/cholmod_super_numeric.c:235-248).  CPU path here; the GPU path in tests/test_gpu_edge_and_demo.py."""
import ctypes as C
import os

import numpy as np
import pytest

from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FA_PATH = os.path.join(ROOT, "tests", "faultalloc", "libfaultalloc.so")


class SSConfig(C.Structure):
    _fields_ = [("malloc_func", C.c_void_p), ("calloc_func", C.c_void_p), ("realloc_func", C.c_void_p),
                ("free_func", C.c_void_p), ("printf_func", C.c_void_p), ("hypot_func", C.c_void_p),
                ("divcomplex_func", C.c_void_p)]


class FaultAllocator:
    """Installs tests/faultalloc's allocator in the library's SuiteSparse_config; restores libc's on exit."""

    def __init__(self, lib):
        self.lib = lib
        self.fa = C.CDLL(FA_PATH)
        self.fa.fa_arm.argtypes = [C.c_long]
        self.fa.fa_calls.restype = C.c_long
        self.fa.fa_failed.restype = C.c_long
        self.cfg = SSConfig.in_dll(lib, "SuiteSparse_config")

    def __enter__(self):
        addr = lambda f: C.cast(f, C.c_void_p).value
        self.cfg.malloc_func, self.cfg.calloc_func = addr(self.fa.fa_malloc), addr(self.fa.fa_calloc)
        self.cfg.realloc_func, self.cfg.free_func = addr(self.fa.fa_realloc), addr(self.fa.fa_free)
        self.fa.fa_arm(-1)
        return self

    def __exit__(self, *a):
        self.fa.fa_arm(-1)
        self.lib.SuiteSparse_start()

    def arm(self, k):
        self.fa.fa_arm(k)

    @property
    def failed(self):
        return self.fa.fa_failed() > 0

    @property
    def calls(self):
        return self.fa.fa_calls()


def run_once(S, n, Ap, Ai, Ax, stype, perm, b):
    """analyze -> factorize -> solve through the library; returns (stage reached, residual or None) and frees whatever
    it got.  Every library call may fail (NULL / FALSE): that ends the run at its stage."""
    cm = C.byref(S.cm)
    nz = int(Ap[-1])
    A = S.L.cholmod_l_allocate_sparse(n, n, max(nz, 1), 1, 1, stype, ch.REAL, cm)
    if not A:
        return "sparse", None
    a = A.contents
    ch._view(a.p, n + 1, C.c_int64, np.int64)[:] = Ap
    ch._view(a.i, nz, C.c_int64, np.int64)[:] = Ai
    ch._view(a.x, nz, C.c_double, np.float64)[:] = Ax
    stage, res, Lf, B, X = "analyze", None, None, None, None
    X2, Y2, E2, Xs = C.POINTER(ch.Dense)(), C.POINTER(ch.Dense)(), C.POINTER(ch.Dense)(), C.POINTER(ch.Sparse)()
    try:
        if perm is None:
            Lf = S.L.cholmod_l_analyze(A, cm)
        else:
            pp = np.ascontiguousarray(perm, dtype=np.int64)
            Lf = S.L.cholmod_l_analyze_p(A, pp.ctypes.data, None, 0, cm)
        if not Lf:
            return stage, None
        stage = "factorize"
        assert Lf.contents.xtype == ch.PATTERN and not Lf.contents.x
        if not S.L.cholmod_l_factorize(A, Lf, cm):
            # L must come back symbolic (cholmod_super_numeric.c:235-248)
            assert Lf.contents.xtype == ch.PATTERN and not Lf.contents.x, (Lf.contents.xtype, Lf.contents.x)
            return stage, None
        stage = "solve"
        B = S.L.cholmod_l_allocate_dense(n, 1, n, ch.REAL, cm)
        if not B:
            return stage, None
        ch._view(B.contents.x, n, C.c_double, np.float64)[:] = b
        X = S.L.cholmod_l_solve(ch.SYS_A, Lf, B, cm)
        if not X:
            return stage, None
        x = ch._view(X.contents.x, n, C.c_double, np.float64).copy()
        res = float(np.linalg.norm(G.sym_matvec(n, Ap, Ai, Ax, stype, x) - b) / np.linalg.norm(b))
        # ... and the solve with a sparse right-hand side (Bset): IPerm, the column -> supernode map, Xset, Y and X are
        # allocated on the way, each of them may fail
        stage = "subset"
        Bs = S.L.cholmod_l_allocate_sparse(n, 1, 2, 0, 1, 0, ch.PATTERN, cm)
        if not Bs:
            return stage, None
        try:
            ch._view(Bs.contents.p, 2, C.c_int64, np.int64)[:] = [0, 2]
            ch._view(Bs.contents.i, 2, C.c_int64, np.int64)[:] = [0, n // 2]
            b2 = np.zeros(n); b2[[0, n // 2]] = [1.0, -2.0]
            ch._view(B.contents.x, n, C.c_double, np.float64)[:] = b2
            ok = S.L.cholmod_l_solve2(ch.SYS_A, Lf, B, Bs, C.byref(X2), C.byref(Xs), C.byref(Y2), C.byref(E2), cm)
            if not ok:
                return stage, None
            k = int(ch._view(Xs.contents.p, 2, C.c_int64, np.int64)[1])
            xset = ch._view(Xs.contents.i, k, C.c_int64, np.int64).copy()
            x2 = ch._view(X2.contents.x, n, C.c_double, np.float64).copy()
            # (X is defined on Xset only: compared with a full solve of the same right-hand side)
            Xf = S.L.cholmod_l_solve(ch.SYS_A, Lf, B, cm)
            if not Xf:
                return stage, None
            xf = ch._view(Xf.contents.x, n, C.c_double, np.float64).copy()
            S.free_dense(Xf)
            assert np.allclose(x2[xset], xf[xset], rtol=1e-11, atol=1e-13)
        finally:
            S.free_sparse(Bs)
        return "done", res
    finally:
        for h, fr in ((X2, S.free_dense), (Y2, S.free_dense), (Xs, S.free_sparse)):
            if h:
                fr(h)
        if X:
            S.free_dense(X)
        if B:
            S.free_dense(B)
        if Lf:
            S.free_factor(Lf)
        S.free_sparse(A)


@pytest.mark.parametrize("case", ["p3d_6", "bcsstk01"])
def test_every_host_allocation_may_fail_cpu_path(case, golden_dir):
    if case == "p3d_6":
        n, Ap, Ai, Ax = G.poisson3d(6)
        stype, perm = -1, G.geometric_nd(6, 6, 6, 3)
    else:
        n, Ap, Ai, Ax, stype = G.read_triplet(os.path.join(golden_dir, "bcsstk01.tri"))
        perm = None
    b = G.demo_rhs(n)
    S = ch.Session(use_gpu=0, ordering="default")
    S.cm.error_handler = ch.ERRFUNC(0)
    count0, inuse0 = S.cm.malloc_count, S.cm.memory_inuse
    stages = set()
    with FaultAllocator(S.L) as fa:
        k, done = 0, False
        while not done:
            assert k < 5000, "the fault loop does not terminate"
            S.cm.status = ch.OK
            fa.arm(k)
            stage, res = run_once(S, n, Ap, Ai, Ax, stype, perm, b)
            failed = fa.failed
            fa.arm(-1)
            assert S.cm.malloc_count == count0 and S.cm.memory_inuse == inuse0, (k, stage, S.cm.malloc_count, S.cm.memory_inuse)
            if failed:
                assert stage != "done" and S.cm.status == ch.OUT_OF_MEMORY, (k, stage, S.cm.status)
                stages.add(stage)
            else:
                assert stage == "done" and res < 1e-11, (k, stage, res)
                done = True
            k += 1
    assert k > 20 and {"analyze", "factorize", "solve", "subset"} <= stages, (k, stages)
    S.finish()


def test_suitesparse_config_entry_points():
    """SuiteSparse_malloc / _calloc / _realloc / _free as the reference defines them (SuiteSparse_config.c:57-330): sizes below
    one are raised to one, a failed growth returns the old block with ok = 0, a failed shrink counts as done."""
    L = ch.lib()
    for f in ("SuiteSparse_malloc", "SuiteSparse_calloc"):
        getattr(L, f).restype = C.c_void_p
        getattr(L, f).argtypes = [C.c_size_t, C.c_size_t]
    L.SuiteSparse_realloc.restype = C.c_void_p
    L.SuiteSparse_realloc.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]
    L.SuiteSparse_free.restype = C.c_void_p
    L.SuiteSparse_free.argtypes = [C.c_void_p]
    p = L.SuiteSparse_calloc(0, 0)
    assert p
    ok = C.c_int(-1)
    p2 = L.SuiteSparse_realloc(100, 1, 8, p, C.byref(ok))
    assert p2 and ok.value == 1
    with FaultAllocator(L) as fa:
        fa.arm(0)
        p3 = L.SuiteSparse_realloc(1000, 100, 8, p2, C.byref(ok))
        assert p3 == p2 and ok.value == 0            # growth failed: old block, ok = 0
        fa.arm(0)
        p4 = L.SuiteSparse_realloc(10, 100, 8, p2, C.byref(ok))
        assert p4 == p2 and ok.value == 1            # a failed shrink leaves the larger block and counts as done
        fa.arm(0)
        assert not L.SuiteSparse_malloc(10, 8)
    assert L.SuiteSparse_free(p2) is None
    assert not L.SuiteSparse_malloc(2 ** 62, 8)     # too large: NULL, no call


def test_rcond_and_change_factor_cpu_path():
    """cholmod_l_rcond (CHOLMOD/Cholesky/cholmod_rcond.c:64-161) and cholmod_l_change_factor's supernodal conversions
    (CHOLMOD/Core/cholmod_change_factor.c:373-393, :946-989) on the CPU path, against numpy on the dense matrix."""
    n, Ap, Ai, Ax = G.poisson3d(5)
    perm = G.geometric_nd(5, 5, 5, 3)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    S.cm.error_handler = ch.ERRFUNC(0)
    assert S.L.cholmod_l_rcond(Lf, C.byref(S.cm)) == -1 and S.cm.status == ch.INVALID      # symbolic L: no values
    assert S.factorize(A, Lf) == 1
    dense = np.zeros((n, n))
    for j in range(n):
        for p in range(Ap[j], Ap[j + 1]):
            dense[Ai[p], j] = dense[j, Ai[p]] = Ax[p]
    Ld = np.linalg.cholesky(dense[np.ix_(ch.FactorView(Lf).Perm, ch.FactorView(Lf).Perm)])
    d = np.diag(Ld)
    rc = S.L.cholmod_l_rcond(Lf, C.byref(S.cm))
    assert abs(rc - (d.min() / d.max()) ** 2) < 1e-13 * rc
    # numeric -> symbolic: values gone, pattern kept; -> numeric: L->x allocated; factorize again
    count = S.cm.malloc_count
    assert S.L.cholmod_l_change_factor(ch.PATTERN, 1, 1, 1, 1, Lf, C.byref(S.cm)) == 1
    f = Lf.contents
    assert f.xtype == ch.PATTERN and not f.x and f.is_super and f.minor == n and S.cm.malloc_count == count - 1
    assert S.L.cholmod_l_check_factor(Lf, C.byref(S.cm)) == 1
    assert S.L.cholmod_l_change_factor(ch.REAL, 1, 1, 1, 1, Lf, C.byref(S.cm)) == 1
    assert Lf.contents.xtype == ch.REAL and Lf.contents.x and S.cm.malloc_count == count
    assert S.L.cholmod_l_change_factor(ch.REAL, 1, 1, 1, 1, Lf, C.byref(S.cm)) == 1          # already numeric: nothing
    assert S.factorize(A, Lf) == 1
    assert abs(S.L.cholmod_l_rcond(Lf, C.byref(S.cm)) - rc) < 1e-13 * rc
    # what is not built says so
    assert S.L.cholmod_l_change_factor(ch.REAL, 1, 0, 1, 1, Lf, C.byref(S.cm)) == 0 and S.cm.status == ch.NOT_INSTALLED
    assert S.L.cholmod_l_change_factor(ch.ZOMPLEX, 1, 1, 1, 1, Lf, C.byref(S.cm)) == 0 and S.cm.status == ch.INVALID
    # a failed factorization: rcond = 0 (L->minor < n)
    Ax2 = Ax.copy()
    Ax2[Ap[int(ch.FactorView(Lf).Perm[n // 2])]] = -1.0
    A2 = S.sparse(n, Ap, Ai, Ax2, -1)
    assert S.factorize(A2, Lf) == 1 and S.cm.status == ch.NOT_POSDEF
    assert S.L.cholmod_l_rcond(Lf, C.byref(S.cm)) == 0
    S.free_sparse(A2)
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()
