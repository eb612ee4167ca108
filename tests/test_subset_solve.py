"""cholmod_l_solve2 with a sparse right-hand side (Bset) on the supernodal factor (reference: Cholesky/cholmod_solve.c:1146-1520,
cholmod_rowfac.c:359-545).  No reference build exists here, so the checks are the properties the reference's algorithm has:
Xset is the reach of P Bset in the elimination tree of L as stored (computed independently below, from the supernodal
pattern), in topological order; X on Xset equals the full solve of the same system with b zero outside Bset; entries of X
outside Xset are not touched; X / Xset / Y handles are reused across calls."""
import ctypes as C

import numpy as np
import pytest

from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G


def _factor(S, n, Ap, Ai, Ax, perm):
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1
    return A, Lf


def _reach(fv, start):
    """the indices reachable from `start` (new numbering) through parent(j) = the first row below the diagonal of column j"""
    sup, pi, s = fv.super, fv.pi, fv.s
    col_super = np.repeat(np.arange(fv.nsuper), np.diff(sup))
    seen = set()
    for i in start:
        while i >= 0 and i not in seen:
            seen.add(int(i))
            q = col_super[i]
            if i + 1 < sup[q + 1]:
                i = i + 1
            else:
                nscol, nsrow = sup[q + 1] - sup[q], pi[q + 1] - pi[q]
                i = int(s[pi[q] + nscol]) if nsrow > nscol else -1
    return seen


@pytest.mark.parametrize("case", ["p2d", "p3d", "natural"])
def test_subset_solve_matches_the_full_solve_on_the_reach(case):
    rng = np.random.default_rng(5)
    if case == "p2d":
        n, Ap, Ai, Ax = G.poisson2d(30); perm = G.geometric_nd(30, 30, 1, 4)
    elif case == "p3d":
        n, Ap, Ai, Ax = G.poisson3d(12); perm = G.geometric_nd(12, 12, 12, 3)
    else:
        n, Ap, Ai, Ax = G.poisson2d(14); perm = np.arange(n)
    S = ch.Session(use_gpu=0)
    A, Lf = _factor(S, n, Ap, Ai, Ax, perm)
    fv = ch.FactorView(Lf)
    P = np.asarray(fv.Perm)
    iperm = np.empty(n, dtype=np.int64); iperm[P] = np.arange(n)
    handles = {}
    for trial, sys in enumerate([ch.SYS_A, ch.SYS_A, ch.SYS_L, ch.SYS_Lt, ch.SYS_LDLt, ch.SYS_LD, ch.SYS_DLt, ch.SYS_D, ch.SYS_P, ch.SYS_Pt]):
        bset = rng.choice(n, size=[1, 3, 7][trial % 3], replace=False)
        b = np.zeros(n); b[bset] = rng.standard_normal(len(bset))
        junk = rng.standard_normal(n)                       # entries of b outside Bset are never read
        bj = junk.copy(); bj[bset] = b[bset]
        x, xset = S.solve_subset(Lf, bj, bset, sys=sys, handles=handles)
        full = S.solve(Lf, b, sys=sys)
        assert len(set(xset.tolist())) == len(xset)
        if sys in (ch.SYS_D, ch.SYS_P, ch.SYS_Pt):
            want = set((iperm[bset] if sys == ch.SYS_P else bset).tolist()) if sys != ch.SYS_Pt else set(P[bset].tolist())
        else:
            start = iperm[bset] if sys == ch.SYS_A else bset
            r = _reach(fv, start.tolist())
            want = set(P[sorted(r)].tolist()) if sys == ch.SYS_A else r
            # topological order: a column comes before every column its pattern reaches
            order = {int(j): k for k, j in enumerate((iperm[xset] if sys == ch.SYS_A else xset).tolist())}
            for j in order:
                for i in _reach(fv, [j]):
                    assert order[i] >= order[j]
        assert set(xset.tolist()) == want, (sys, sorted(want)[:10], sorted(xset.tolist())[:10])
        assert np.allclose(x[xset], full[xset], rtol=1e-12, atol=1e-13), sys
        if sys in (ch.SYS_A, ch.SYS_L, ch.SYS_LD, ch.SYS_LDLt):
            # the forward solve's result is exactly zero outside the reach, so for these systems the subset is not a cut
            pass
        if sys == ch.SYS_L:
            assert np.all(full[np.setdiff1d(np.arange(n), xset)] == 0)
    S.free_subset_handles(handles)
    f = Lf.contents
    assert f.is_super == 1 and f.bset_work and (case == "natural" or f.IPerm)      # L stays supernodal; the caches live in L
    S.free_factor(Lf); S.free_sparse(A); S.finish()
    assert S.cm.malloc_count == 0 if hasattr(S.cm, "malloc_count") else True


def test_subset_solve_with_an_index_listed_many_times():
    """A Bset longer than n (every index several times): Xset holds each index once for every system -- the P / Pt / D
    systems copied Bset as it came, into n slots (round-5 advisor)."""
    n, Ap, Ai, Ax = G.poisson2d(6)
    perm = G.geometric_nd(6, 6, 1, 2)
    S = ch.Session(use_gpu=0)
    A, Lf = _factor(S, n, Ap, Ai, Ax, perm)
    rng = np.random.default_rng(2)
    base = rng.choice(n, size=5, replace=False)
    bset = np.concatenate([base] * 9)                   # 45 entries for n = 36
    assert len(bset) > n
    b = np.zeros(n); b[base] = rng.standard_normal(len(base))
    for sys in (ch.SYS_P, ch.SYS_Pt, ch.SYS_D, ch.SYS_A, ch.SYS_L):
        x, xset = S.solve_subset(Lf, b, bset, sys=sys)
        assert len(set(xset.tolist())) == len(xset) <= n, sys
        full = S.solve(Lf, b, sys=sys)
        assert np.allclose(x[xset], full[xset], rtol=1e-12, atol=1e-13), sys
    S.free_factor(Lf); S.free_sparse(A); S.finish()
    assert S.cm.malloc_count == 0


def test_subset_solve_of_a_complex_factor():
    n, Ap, Ai, Ax = G.poisson2d(16)
    rng = np.random.default_rng(2)
    # Hermitian: random phases on the off-diagonal entries of the lower triangle, diagonal kept real and dominant
    col = np.repeat(np.arange(n), np.diff(Ap))
    ph = np.exp(1j * rng.uniform(0, 2 * np.pi, len(Ax)))
    Az = np.where(Ai == col, Ax + 0j, Ax * ph)
    S = ch.Session(use_gpu=0)
    A, Lf = _factor(S, n, Ap, Ai, Az, G.geometric_nd(16, 16, 1, 4))
    bset = np.array([3, 77, 200])
    b = np.zeros(n, dtype=complex); b[bset] = rng.standard_normal(3) + 1j * rng.standard_normal(3)
    for sys in (ch.SYS_A, ch.SYS_L, ch.SYS_Lt):
        x, xset = S.solve_subset(Lf, b, bset, sys=sys)
        full = S.solve(Lf, b, sys=sys)
        assert np.allclose(x[xset], full[xset], rtol=1e-12, atol=1e-13), sys
    S.free_factor(Lf); S.free_sparse(A); S.finish()


def test_subset_solve_argument_errors():
    n, Ap, Ai, Ax = G.poisson2d(8)
    S = ch.Session(use_gpu=0)
    A, Lf = _factor(S, n, Ap, Ai, Ax, np.arange(n))
    with pytest.raises(RuntimeError):
        S.solve_subset(Lf, np.ones(n), np.array([n + 3]))          # index outside 0 .. n-1
    assert S.cm.status == ch.INVALID
    # two right-hand sides / a complex B against a real L: CHOLMOD_INVALID as in the reference (cholmod_solve.c:1081-1094)
    B2 = S.dense(np.ones((2, n)))
    Bs = S.L.cholmod_l_allocate_sparse(n, 1, 1, 0, 1, 0, ch.PATTERN, C.byref(S.cm))
    X = C.POINTER(ch.Dense)(); Xs = C.POINTER(ch.Sparse)(); Y = C.POINTER(ch.Dense)(); E = C.POINTER(ch.Dense)()
    assert S.L.cholmod_l_solve2(ch.SYS_A, Lf, B2, Bs, C.byref(X), C.byref(Xs), C.byref(Y), C.byref(E), C.byref(S.cm)) == 0
    assert S.cm.status == ch.INVALID
    Bc = S.dense(np.ones(n) + 1j)
    assert S.L.cholmod_l_solve2(ch.SYS_A, Lf, Bc, Bs, C.byref(X), C.byref(Xs), C.byref(Y), C.byref(E), C.byref(S.cm)) == 0
    assert S.cm.status == ch.INVALID
    S.free_dense(B2); S.free_dense(Bc); S.free_sparse(Bs)
    if X: S.free_dense(X)
    S.free_factor(Lf); S.free_sparse(A); S.finish()


@pytest.mark.gpu
@pytest.mark.parametrize("on_device", [False, True])
def test_subset_solve_of_a_factor_computed_on_the_gpu(on_device):
    """The subset solve reads L on the host: a factor that lives in HBM only is downloaded once (and stays resident for
    the full solves, which run on the device)."""
    n, Ap, Ai, Ax = G.poisson3d(20)
    S = ch.Session(factor_on_device=on_device)
    A, Lf = _factor(S, n, Ap, Ai, Ax, G.geometric_nd(20, 20, 20, 4))
    rng = np.random.default_rng(9)
    h = {}
    for sys in (ch.SYS_A, ch.SYS_L, ch.SYS_Lt, ch.SYS_A):
        bset = rng.choice(n, size=5, replace=False)
        b = np.zeros(n); b[bset] = rng.standard_normal(5)
        x, xset = S.solve_subset(Lf, b, bset, sys=sys, handles=h)
        full = S.solve(Lf, b, sys=sys)
        assert 0 < len(xset) < n
        assert np.allclose(x[xset], full[xset], rtol=1e-11, atol=1e-13), sys
    # a new factorization of the same matrix scaled by 4: the host copy is refreshed, not reused
    A2 = S.sparse(n, Ap, Ai, 4.0 * Ax, -1)
    assert S.factorize(A2, Lf) == 1
    bset = np.array([0, n // 2]); b = np.zeros(n); b[bset] = [1.0, -2.0]
    x, xset = S.solve_subset(Lf, b, bset, handles=h)
    full = S.solve(Lf, b)
    assert np.allclose(x[xset], full[xset], rtol=1e-11, atol=1e-13)
    S.free_subset_handles(h)
    S.free_factor(Lf); S.free_sparse(A); S.free_sparse(A2); S.finish()
