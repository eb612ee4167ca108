"""The engine's plan is built inside cholmod_l_analyze (round 6; Common->hip_lazy_plan = 0) -- the reference cuts its device
pools inside the analysis too (CHOLMOD/Supernodal/cholmod_super_symbolic.c:243-327) -- so that the first cholmod_l_factorize of
a symbolic factor costs a factorization, not a factorization plus schedule, maps and the reservation of HBM."""
import ctypes as C

import numpy as np
import pytest

from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G


def _problem(m=10):
    n, Ap, Ai, Ax = G.poisson3d(m)
    perm = G.geometric_nd(m, m, m, 3)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    assert O.factorize(Ax) == 0
    return n, Ap, Ai, Ax, perm, O


def _check(S, A, Lf, O, n, Ap, Ai, Ax):
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    for key in ("super", "pi", "px", "s", "Perm"):
        assert np.array_equal(getattr(fv, key), getattr(O, key)), key
    m = O.lower_mask()
    assert np.linalg.norm((fv.x - O.x)[m]) <= 1e-12 * np.linalg.norm(O.x[m])
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    assert np.linalg.norm(G.sym_matvec(n, Ap, Ai, Ax, -1, x) - b) <= 1e-11 * np.linalg.norm(b)


def test_without_a_device_the_analysis_builds_no_plan_and_says_nothing():
    """(runs everywhere; on a GPU box it is the Common->useGPU = 0 case)"""
    n, Ap, Ai, Ax, perm, O = _problem(6)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.cm.status == ch.OK and not Lf.contents.hip_plan and Lf.contents.hip_plan_ahead == 0
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


@pytest.mark.gpu
def test_analysis_leaves_a_plan_and_the_first_factorization_uses_it():
    n, Ap, Ai, Ax, perm, O = _problem()
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert Lf.contents.useGPU == 1 and Lf.contents.hip_plan and Lf.contents.hip_plan_ahead != 0
    assert S.cm.hip_plan_seconds > 0
    plan = Lf.contents.hip_plan
    S.cm.hip_plan_seconds = -1.0
    _check(S, A, Lf, O, n, Ap, Ai, Ax)
    assert Lf.contents.hip_plan == plan and Lf.contents.hip_plan_ahead == 0     # the same plan, now in use
    assert S.cm.hip_plan_seconds == -1.0                                        # ... nothing was built again
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


@pytest.mark.gpu
def test_lazy_plan_on_request():
    n, Ap, Ai, Ax, perm, O = _problem()
    S = ch.Session()
    S.cm.hip_lazy_plan = 1
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert Lf.contents.useGPU == 1 and not Lf.contents.hip_plan
    _check(S, A, Lf, O, n, Ap, Ai, Ax)
    assert Lf.contents.hip_plan and S.cm.hip_plan_seconds > 0
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


@pytest.mark.gpu
def test_plan_flags_changed_after_the_analysis_rebuild_the_plan():
    """Common->hip_flags belongs to the factorization: a plan built ahead with other flags is replaced, not used."""
    n, Ap, Ai, Ax, perm, O = _problem(16)
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert Lf.contents.hip_plan and Lf.contents.hip_plan_ahead == 1             # flags 0, + 1
    S.cm.hip_flags = 1024 | 512        # no fused solve + update launches, every diagonal block by a dpotrf launch of its own
    S.cm.hip_plan_seconds = -1.0
    _check(S, A, Lf, O, n, Ap, Ai, Ax)
    assert S.cm.hip_plan_seconds > 0                                            # built again
    st = S.hip_stats(Lf)
    assert st[31] == 0 and st[26] == 0                                          # no k_trsm_upd, no k_update2f launches
    # and a second factorization keeps it
    S.cm.hip_plan_seconds = -1.0
    assert S.factorize(A, Lf) == 1 and S.cm.hip_plan_seconds == -1.0
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


@pytest.mark.gpu
def test_a_complex_matrix_gets_its_plan_with_the_twin():
    n, Ap, Ai, Ax, perm, O = _problem(6)
    Az = Ax.astype(np.complex128)
    Az[Ai != np.repeat(np.arange(n), np.diff(Ap))] *= (1.0 + 0.25j) / abs(1.0 + 0.25j)
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Az, -1)
    Lf = S.analyze(A, perm)
    assert not Lf.contents.hip_plan and not Lf.contents.cx_twin
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    assert Lf.contents.cx_twin
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


@pytest.mark.gpu
def test_a_failed_reservation_inside_the_analysis_is_reported_by_the_factorization(monkeypatch):
    n, Ap, Ai, Ax, perm, O = _problem(8)
    monkeypatch.setenv("CHOLMOD_HIP_TEST_FAIL_ALLOC", "1")
    S = ch.Session(hooks=True)
    seen = []
    handler = ch.ERRFUNC(lambda st, f, l, m: seen.append(st))
    S.cm.error_handler = handler
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.cm.status == ch.OK and Lf.contents.useGPU == 1 and not Lf.contents.hip_plan and seen == []
    assert S.factorize(A, Lf) == 0 and S.cm.status == ch.OUT_OF_MEMORY and seen and seen[-1] == ch.OUT_OF_MEMORY
    monkeypatch.delenv("CHOLMOD_HIP_TEST_FAIL_ALLOC")
    _check(S, A, Lf, O, n, Ap, Ai, Ax)
    S.cm.error_handler = ch.ERRFUNC()
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


@pytest.mark.gpu
def test_a_complex_factorization_after_a_real_analysis_drops_the_unused_plan():
    """analyze sees a real matrix (plan built ahead), factorize gets Hermitian values on the same pattern: the twin factor
    brings its own plan, the unused one must not keep its HBM"""
    n, Ap, Ai, Ax, perm, O = _problem(6)
    Az = Ax.astype(np.complex128)
    off = Ai != np.repeat(np.arange(n), np.diff(Ap))
    Az[off] *= np.exp(0.3j)
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert Lf.contents.hip_plan and Lf.contents.hip_plan_ahead != 0
    Ac = S.sparse(n, Ap, Ai, Az, -1)
    assert S.factorize(Ac, Lf) == 1 and S.cm.status == ch.OK
    assert Lf.contents.cx_twin and not Lf.contents.hip_plan and Lf.contents.hip_plan_ahead == 0
    b = G.demo_rhs(n).astype(np.complex128)
    x = S.solve(Lf, b)
    import scipy.sparse as sp
    Lo = sp.csc_matrix((Az, Ai, Ap), shape=(n, n))
    Af = Lo + sp.tril(Lo, -1).conj().T
    assert np.linalg.norm(Af @ x - b) <= 1e-11 * np.linalg.norm(b)
    S.free_factor(Lf)
    S.free_sparse(A)
    S.free_sparse(Ac)
    assert S.cm.malloc_count == 0
    S.finish()
