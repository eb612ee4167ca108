"""Edge cases of the path (as the reference's Tcov torture matrices probe them:
empty, 1x1, diagonal, dense, unpacked / unsorted / upper-stored inputs) and the
C demo driver that mirrors CHOLMOD/Demo/cholmod_l_demo.c on bcsstk01."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(n, Ap, Ai, Ax, stype, perm=None, tol=1e-12):
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    assert O.factorize(Ax) == 0
    for k in ("Perm", "super", "pi", "px", "s"):
        assert np.array_equal(getattr(fv, k), getattr(O, k)), k
    if n:
        m = O.lower_mask()
        assert np.linalg.norm((fv.x - O.x)[m]) <= tol * np.linalg.norm(O.x[m])
        b = G.demo_rhs(n)
        x = S.solve(Lf, b)
        r = G.sym_matvec(n, Ap, Ai, Ax, stype, x) - b
        assert np.linalg.norm(r) <= 1e-11 * np.linalg.norm(b)
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


def test_empty_matrix():
    _check(0, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0), -1)


def test_one_by_one():
    _check(1, np.array([0, 1]), np.array([0]), np.array([4.0]), -1)


def test_diagonal_matrix():
    n = 37
    _check(n, np.arange(n + 1), np.arange(n), np.linspace(1.0, 9.0, n), -1)


@pytest.mark.parametrize("n", [50, 200, 700, 2600])
def test_dense_spd_single_supernode(n):
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    Ad = M @ M.T + n * np.eye(n)
    ii, jj = np.tril_indices(n)
    order = np.lexsort((ii, jj))
    Ai, cols, Ax = ii[order], jj[order], Ad[ii[order], jj[order]]
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols, minlength=n), out=Ap[1:])
    _check(n, Ap, Ai.astype(np.int64), Ax, -1)


def test_arrow_matrix_with_given_perm():
    n = 300
    rows = np.concatenate([np.arange(n), np.full(n - 1, n - 1)])
    cols = np.concatenate([np.arange(n), np.arange(n - 1)])
    vals = np.concatenate([np.full(n, float(n)), np.full(n - 1, -1.0)])
    order = np.lexsort((rows, cols))
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols, minlength=n), out=Ap[1:])
    _check(n, Ap, rows[order].astype(np.int64), vals[order], -1, perm=np.arange(n)[::-1].copy())


def test_upper_stored_and_unpacked_unsorted_input():
    """stype > 0 goes through one ptranspose, an unpacked matrix with unsorted
    columns (nz[] used, gaps in i/x) through the unpacked code paths."""
    n, Ap, Ai, Ax = G.poisson3d(9)
    perm = G.geometric_nd(9, 9, 9, 3)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    assert O.factorize(Ax) == 0
    m = O.lower_mask()
    S = ch.Session()
    # upper: transpose of the lower CSC = CSR of lower = CSC of upper
    A_lo = S.sparse(n, Ap, Ai, Ax, -1)
    A_up = S.L.cholmod_l_ptranspose(A_lo, 2, None, None, 0, C.byref(S.cm))
    assert A_up.contents.stype == 1
    # unpacked + unsorted copy of the lower matrix
    cap = int(Ap[-1]) + 3 * n
    A_un = S.L.cholmod_l_allocate_sparse(n, n, cap, 0, 0, -1, ch.REAL, C.byref(S.cm))
    a = A_un.contents
    p = ch._view(a.p, n + 1, C.c_int64, np.int64)
    nz = ch._view(a.nz, n, C.c_int64, np.int64)
    ai = ch._view(a.i, cap, C.c_int64, np.int64)
    ax = ch._view(a.x, cap, C.c_double, np.float64)
    ai[:] = 0
    ax[:] = np.nan
    rng = np.random.default_rng(1)
    pos = 0
    for j in range(n):
        cnt = int(Ap[j + 1] - Ap[j])
        o = rng.permutation(cnt)
        p[j] = pos
        nz[j] = cnt
        ai[pos:pos + cnt] = Ai[Ap[j]:Ap[j + 1]][o]
        ax[pos:pos + cnt] = Ax[Ap[j]:Ap[j + 1]][o]
        pos += cnt + 3
    p[n] = pos
    for A in (A_up, A_un):
        Lf = S.analyze(A, perm)
        assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
        fv = ch.FactorView(Lf)
        assert np.array_equal(fv.s, O.s) and np.array_equal(fv.px, O.px)
        assert np.linalg.norm((fv.x - O.x)[m]) <= 1e-12 * np.linalg.norm(O.x[m])
        S.free_factor(Lf)
    for A in (A_lo, A_up, A_un):
        S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


def test_every_host_allocation_may_fail_gpu_path():
    """The Tcov-shaped fault loop of tests/test_memory_faults.py on the GPU path: the k-th allocation of the host C layer
    fails, k = 0, 1, ... through analyze -> plan build -> factorize (engine) -> solve (device).  After every run
    Common->malloc_count / memory_inuse are back where they were, a failed run reports CHOLMOD_OUT_OF_MEMORY, and a
    factorization that fails on a symbolic L hands it back symbolic (CHOLMOD/Supernodal/cholmod_super_numeric.c:235-248)."""
    from test_memory_faults import FaultAllocator, run_once
    n, Ap, Ai, Ax = G.poisson3d(7)
    perm = G.geometric_nd(7, 7, 7, 3)
    b = G.demo_rhs(n)
    S = ch.Session(use_gpu=1, ordering="default")
    S.cm.error_handler = ch.ERRFUNC(0)
    # (once without faults: device context, kernels loaded)
    stage, res = run_once(S, n, Ap, Ai, Ax, -1, perm, b)
    assert stage == "done" and res < 1e-11
    count0, inuse0 = S.cm.malloc_count, S.cm.memory_inuse
    stages = set()
    with FaultAllocator(S.L) as fa:
        k, done = 0, False
        while not done:
            assert k < 5000
            S.cm.status = ch.OK
            fa.arm(k)
            stage, res = run_once(S, n, Ap, Ai, Ax, -1, perm, b)
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
    assert k > 20 and {"analyze", "factorize", "solve"} <= stages, (k, stages)
    S.finish()


def test_rcond_and_change_factor_on_the_device(golden_dir):
    """cholmod_l_rcond on a factor that lives in HBM (one pass over the diagonal on the device, nothing downloaded:
    CHOLMOD/Cholesky/cholmod_rcond.c:64-161 restated) against the CPU path's value; cholmod_l_change_factor numeric ->
    symbolic forgets the resident values (the plan stays), the next factorization brings them back."""
    for case in ("p3d", "bcsstk01"):
        if case == "p3d":
            n, Ap, Ai, Ax = G.poisson3d(12)
            stype, perm = -1, G.geometric_nd(12, 12, 12, 3)
        else:
            n, Ap, Ai, Ax, stype = G.read_triplet(os.path.join(golden_dir, "bcsstk01.tri"))
            perm = None
        rc = {}
        for use_gpu, on_device in ((0, False), (1, False), (1, True)):
            S = ch.Session(use_gpu=use_gpu, factor_on_device=on_device, ordering="default")
            A = S.sparse(n, Ap, Ai, Ax, stype)
            Lf = S.analyze(A, perm)
            assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
            if on_device:
                assert not Lf.contents.hip_host_valid
            rc[(use_gpu, on_device)] = S.L.cholmod_l_rcond(Lf, C.byref(S.cm))
            if on_device:
                assert not Lf.contents.hip_host_valid              # rcond did not download the factor
                assert S.L.cholmod_l_change_factor(ch.PATTERN, 1, 1, 1, 1, Lf, C.byref(S.cm)) == 1
                assert Lf.contents.xtype == ch.PATTERN and not Lf.contents.x and not Lf.contents.hip_on_device
                S.cm.error_handler = ch.ERRFUNC(0)
                assert S.L.cholmod_l_rcond(Lf, C.byref(S.cm)) == -1 and S.cm.status == ch.INVALID
                assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
                assert abs(S.L.cholmod_l_rcond(Lf, C.byref(S.cm)) - rc[(1, True)]) <= 1e-14 * rc[(1, True)]
            S.free_factor(Lf)
            S.free_sparse(A)
            assert S.cm.malloc_count == 0
            S.finish()
        ref = rc[(0, False)]
        assert 0 < ref < 1
        assert abs(rc[(1, False)] - ref) < 1e-12 * ref and abs(rc[(1, True)] - ref) < 1e-12 * ref, rc


def test_device_allocation_failure_is_loud_or_degrades_on_request(monkeypatch):
    """The reservation of L in HBM fails (test hook): by default the factorization
    fails with CHOLMOD_OUT_OF_MEMORY and L stays symbolic
    (CHOLMOD/Supernodal/cholmod_super_numeric.c:235-248); with
    Common->hip_cpu_fallback it degrades to the CPU path with status OK, as the
    reference does when its GPU cannot be initialised (t_cholmod_super_numeric.c:183-192)."""
    n, Ap, Ai, Ax = G.poisson3d(9)
    perm = G.geometric_nd(9, 9, 9, 3)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    assert O.factorize(Ax) == 0
    monkeypatch.setenv("CHOLMOD_HIP_TEST_FAIL_ALLOC", "1")
    S = ch.Session(hooks=True)          # (the engine's test hooks live in lib/libcholmod_amd_testhooks.so only)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert Lf.contents.useGPU == 1
    assert S.factorize(A, Lf) == 0 and S.cm.status == ch.OUT_OF_MEMORY
    assert not Lf.contents.x and Lf.contents.xtype == ch.PATTERN and not Lf.contents.hip_plan
    S.cm.hip_cpu_fallback = 1
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    assert not Lf.contents.hip_plan and S.cm.cholmod_cpu_potrf_calls == fv.nsuper
    m = O.lower_mask()
    assert np.linalg.norm((fv.x - O.x)[m]) <= 1e-12 * np.linalg.norm(O.x[m])
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    assert np.linalg.norm(G.sym_matvec(n, Ap, Ai, Ax, -1, x) - b) <= 1e-11 * np.linalg.norm(b)
    # the device comes back: the same factor object moves to the engine
    monkeypatch.delenv("CHOLMOD_HIP_TEST_FAIL_ALLOC")
    Lf.contents.useGPU = 1
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK and Lf.contents.hip_plan
    assert np.linalg.norm((ch.FactorView(Lf).x - O.x)[m]) <= 1e-12 * np.linalg.norm(O.x[m])
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


def test_use_gpu_zero_runs_the_cpu_path_on_a_gpu_box():
    n, Ap, Ai, Ax = G.poisson3d(8)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, None)
    assert Lf.contents.useGPU == 0
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK and not Lf.contents.hip_plan
    assert S.cm.cholmod_gpu_potrf_calls == 0 and S.cm.cholmod_cpu_potrf_calls == ch.FactorView(Lf).nsuper
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    assert np.linalg.norm(G.sym_matvec(n, Ap, Ai, Ax, -1, x) - b) <= 1e-11 * np.linalg.norm(b)
    assert not Lf.contents.hip_plan                          # the solve stayed on the host too
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_c_demo_driver_on_bcsstk01(golden_dir, tmp_path):
    """BASELINE.json config #1: the demo flow in C against include/cholmod.h."""
    rec = json.load(open(os.path.join(golden_dir, "reference_recorded.json")))["bcsstk01"]
    exe = str(tmp_path / "cholmod_l_demo")
    lib = os.path.join(ROOT, "suitesparse_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cholmod_l_demo.c"), "-L", lib, "-lcholmod_amd",
                           f"-Wl,-rpath,{lib}", "-lm", "-o", exe])
    permfile = tmp_path / "perm.txt"
    permfile.write_text(" ".join(str(v) for v in rec["Perm"]))
    with open(os.path.join(golden_dir, "bcsstk01.tri")) as f:
        out = subprocess.run([exe, "-perm", str(permfile)], stdin=f, capture_output=True, text=True, timeout=300)
    from test_cpu_path import _check_demo_output
    _check_demo_output(out)
    assert "(HIP engine)" in out.stdout and "kernel launches 0" not in out.stdout


def test_big_supernode_block_walk_solves_multi_rhs():
    """A 1400-column supernode with 100 rows below it (> 1024 columns: the solve
    walks it in 64-column blocks with the precomputed diagonal-block inverses,
    k_solve_fwd_blk / k_solve_bwd_blk) followed by a dense 200-column root;
    three right-hand sides, L / L' / A systems against the oracle."""
    n1, n2 = 1400, 200
    n = n1 + n2
    rng = np.random.default_rng(7)
    M = rng.standard_normal((n, n)) * 0.05
    Ad = M @ M.T + np.eye(n) * 4.0
    mask = np.zeros((n, n), dtype=bool)
    mask[:n1, :n1] = True
    mask[n1:, n1:] = True
    mask[n1 + 100:, :n1] = True            # only the last 100 rows couple to the first block
    mask[:n1, n1 + 100:] = True
    Ad = np.where(mask, Ad, 0.0)
    Ad += np.eye(n) * (np.abs(Ad).sum(axis=1).max())
    ii, jj = np.nonzero(np.tril(mask))
    order = np.lexsort((ii, jj))
    Ai, cols = ii[order].astype(np.int64), jj[order]
    Ax = Ad[Ai, cols]
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols, minlength=n), out=Ap[1:])
    perm = np.arange(n, dtype=np.int64)
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    assert O.factorize(Ax) == 0
    assert np.array_equal(fv.super, O.super) and fv.nsuper >= 2
    assert np.diff(fv.super).max() >= 1400 - 64
    m = O.lower_mask()
    assert np.linalg.norm((fv.x - O.x)[m]) <= 1e-12 * np.linalg.norm(O.x[m])
    b = rng.standard_normal((3, n))
    y = S.solve(Lf, b, ch.SYS_L)
    assert np.linalg.norm(y - O.lsolve(b)) / np.linalg.norm(y) < 1e-11
    z = S.solve(Lf, b, ch.SYS_Lt)
    assert np.linalg.norm(z - O.ltsolve(b)) / np.linalg.norm(z) < 1e-11
    x = S.solve(Lf, b)
    for k in range(3):
        r = G.sym_matvec(n, Ap, Ai, Ax, -1, x[k]) - b[k]
        assert np.linalg.norm(r) / np.linalg.norm(b[k]) < 1e-11
    # a second factorization invalidates the inverses: solve again after refactorizing
    assert S.factorize(A, Lf) == 1
    x2 = S.solve(Lf, b[0])
    assert np.linalg.norm(x2 - x[0]) / np.linalg.norm(x[0]) < 1e-12
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


@pytest.mark.parametrize("seed,n,density", [(1, 400, 0.02), (2, 900, 0.008), (3, 1500, 0.004)])
def test_random_sparse_spd(seed, n, density):
    """Unstructured patterns (no grid, no given ordering): natural order plus the
    reference's postorder; maps bit-exact, factor and residual vs the oracle."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    R = sp.random(n, n, density=density, random_state=seed, format="csr")
    Asym = (R + R.T).tocsr()
    Asym.data[:] = -np.abs(Asym.data) - 0.1
    Asym = Asym + sp.diags(np.asarray(-Asym.sum(axis=1)).ravel() + 1.0)
    T = sp.tril(Asym).tocsc()
    T.sort_indices()
    _check(n, T.indptr.astype(np.int64), T.indices.astype(np.int64), T.data.astype(np.float64), -1, tol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("stype", [-1, 1])
def test_factorize_again_with_new_values_and_with_a_new_pattern(stype):
    """cholmod_l_factorize called repeatedly on one L: the same pattern with new values takes
    the values-only path (hash of p / i; H2D of A->x + gather into the resident S), a changed
    pattern the full path again; every result against the oracle."""
    import scipy.sparse as sp
    n, Ap, Ai, Ax = G.poisson3d(14)
    perm = G.geometric_nd(14, 14, 14, 4)
    if stype > 0:                                   # the same matrix, upper triangle stored
        U = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)).T.tocsc()
        U.sort_indices()
        Ap, Ai, Ax = U.indptr.astype(np.int64), U.indices.astype(np.int64), U.data.copy()
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    mask = O.lower_mask()
    rng = np.random.default_rng(5)
    a = A.contents
    for it in range(4):
        vals = Ax * (1.0 + 0.3 * it)
        diag = Ai == np.repeat(np.arange(n), np.diff(Ap))
        vals[diag] += rng.uniform(0.0, 1.0, int(diag.sum()))
        ch._view(a.x, len(vals), C.c_double, np.float64)[:] = vals
        assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
        assert Lf.contents.hip_apat_valid == 1
        assert O.factorize(vals) == 0
        fv = ch.FactorView(Lf)
        assert np.linalg.norm((fv.x - O.x)[mask]) / np.linalg.norm(O.x[mask]) < 1e-12, it
        b = G.demo_rhs(n)
        x = S.solve(Lf, b)
        assert np.linalg.norm(G.sym_matvec(n, Ap, Ai, vals, stype, x) - b) / np.linalg.norm(b) < 1e-11
    # a different pattern (one off-diagonal entry removed everywhere it is stored) on the same L:
    # entries outside the old pattern are not a concern, fewer entries are simply absent
    keep = np.ones(len(Ai), dtype=bool)
    off = np.where(~diag)[0]
    keep[off[len(off) // 2]] = False
    cols = np.repeat(np.arange(n), np.diff(Ap))[keep]
    Ap2 = np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=n))]).astype(np.int64)
    Ai2, Ax2 = Ai[keep].copy(), vals[keep].copy()
    A2 = S.sparse(n, Ap2, Ai2, Ax2, stype)
    h_before = Lf.contents.hip_apat_hash
    assert S.factorize(A2, Lf) == 1 and S.cm.status == ch.OK
    assert Lf.contents.hip_apat_hash != h_before
    O2 = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)      # L's symbolic structure is the old one
    assert O2.factorize(Ax2, Ap=Ap2, Ai=Ai2) == 0
    fv = ch.FactorView(Lf)
    assert np.linalg.norm((fv.x - O2.x)[mask]) / np.linalg.norm(O2.x[mask]) < 1e-12
    # another pattern with the SAME number of entries (one row index moved): the values-only path starts the factorization
    # before the pattern hash is in (round 6) -- the hash must then void what the device computed and the call must take
    # the long way; twice, so that the second call goes through the values-only path of the NEW pattern
    Ai3 = Ai2.copy()
    cols2 = np.repeat(np.arange(n), np.diff(Ap2))
    moved = False
    for q in range(len(Ai3) - 1, 0, -1):
        j = cols2[q]
        lo = Ai3[q - 1] if cols2[q - 1] == j else -1
        cand = Ai3[q] - 1 if stype < 0 else Ai3[q] + 1
        # keep the column sorted, off the diagonal, inside the stored triangle and away from its neighbours
        if stype < 0 and cand > j and cand > lo and Ai3[q] != j:
            Ai3[q] = cand ; moved = True ; break
        if stype > 0 and cand < j and Ai3[q] != j and (q + 1 == len(Ai3) or cols2[q + 1] != j or Ai3[q + 1] > cand):
            Ai3[q] = cand ; moved = True ; break
    assert moved
    A3 = S.sparse(n, Ap2, Ai3, Ax2, stype)
    for _ in range(2):
        h_before = Lf.contents.hip_apat_hash
        assert S.factorize(A3, Lf) == 1 and S.cm.status == ch.OK
        assert O2.factorize(Ax2, Ap=Ap2, Ai=Ai3) == 0
        fv = ch.FactorView(Lf)
        assert np.linalg.norm((fv.x - O2.x)[mask]) / np.linalg.norm(O2.x[mask]) < 1e-12
    assert Lf.contents.hip_apat_hash == h_before            # (second call: same pattern, values only)
    # ... and back to the previous pattern, again with the same count
    assert S.factorize(A2, Lf) == 1 and S.cm.status == ch.OK
    assert O2.factorize(Ax2, Ap=Ap2, Ai=Ai2) == 0
    fv = ch.FactorView(Lf)
    assert np.linalg.norm((fv.x - O2.x)[mask]) / np.linalg.norm(O2.x[mask]) < 1e-12
    S.free_factor(Lf)
    S.free_sparse(A)
    S.free_sparse(A2)
    S.free_sparse(A3)
    assert S.cm.malloc_count == 0
    S.finish()
