"""Built-in nested dissection ordering (suitesparse_amd/csrc/host/order.c): the
stand-in for the ordering packages the reference calls in cholmod_analyze
(Cholesky/cholmod_analyze.c:569-804).  No reference permutation to match: the
tests check that it is a permutation, that it reduces fill, and that the
supernodal maps built on it are the oracle's for that same final Perm."""
import numpy as np
import pytest

from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G


def _analyze(n, Ap, Ai, Ax, ordering, use_gpu=0):
    S = ch.Session(use_gpu=use_gpu, ordering=ordering)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A)
    fv = ch.FactorView(Lf)
    out = dict(Perm=fv.Perm.copy(), super=fv.super.copy(), pi=fv.pi.copy(), px=fv.px.copy(), s=fv.s.copy(),
               lnz=S.cm.lnz, fl=S.cm.fl, ordering=int(Lf.contents.ordering))
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()
    return out


def _random_spd(n, density, seed):
    import scipy.sparse as sp
    R = sp.random(n, n, density=density, random_state=seed, format="csr")
    Asym = (R + R.T).tocsr()
    Asym.data[:] = -np.abs(Asym.data) - 0.1
    Asym = Asym + sp.diags(np.asarray(-Asym.sum(axis=1)).ravel() + 1.0)
    T = sp.tril(Asym).tocsc()
    T.sort_indices()
    return n, T.indptr.astype(np.int64), T.indices.astype(np.int64), T.data.astype(np.float64)


CASES = {
    "p3d_16": lambda: G.poisson3d(16),
    "p2d_70": lambda: G.poisson2d(70),
    "box8": lambda: G.box_stencil3d(8, 2),
    "rand600": lambda: _random_spd(600, 0.01, 5),
    "two_blocks": lambda: _two_blocks(),
}


def _two_blocks():
    # two disconnected grids + isolated vertices
    n1, Ap1, Ai1, Ax1 = G.poisson2d(20)
    n2, Ap2, Ai2, Ax2 = G.poisson2d(13)
    iso = 7
    n = n1 + n2 + iso
    Ap = np.concatenate([Ap1, Ap1[-1] + Ap2[1:], Ap1[-1] + Ap2[-1] + np.arange(1, iso + 1)])
    Ai = np.concatenate([Ai1, Ai2 + n1, n1 + n2 + np.arange(iso)])
    Ax = np.concatenate([Ax1, Ax2, np.full(iso, 2.0)])
    return n, Ap.astype(np.int64), Ai.astype(np.int64), Ax


@pytest.mark.parametrize("name", sorted(CASES))
def test_nested_dissection_is_a_fill_reducing_permutation(name):
    n, Ap, Ai, Ax = CASES[name]()
    nat = _analyze(n, Ap, Ai, Ax, "natural")
    nd = _analyze(n, Ap, Ai, Ax, "nesdis")
    dflt = _analyze(n, Ap, Ai, Ax, "default")
    assert sorted(nd["Perm"].tolist()) == list(range(n))
    assert nd["ordering"] == 4 and nat["ordering"] in (0, 6)
    # cholmod_l_start's default strategy is the built-in dissection; deterministic
    assert np.array_equal(dflt["Perm"], nd["Perm"])
    if name != "box8":          # 512 vertices, 124 neighbours each: nothing to dissect, the band order wins
        assert nd["lnz"] <= nat["lnz"] * (1.0 if name != "rand600" else 1.05)
    if name in ("p3d_16", "p2d_70"):
        assert nd["fl"] < 0.5 * nat["fl"]
    # the maps on that ordering are the oracle's for the same final permutation
    O = OracleFactor(n, Ap, Ai, -1, perm=nd["Perm"], postorder=True)
    for k in ("Perm", "super", "pi", "px", "s"):
        assert np.array_equal(nd[k], getattr(O, k)), k


def test_degenerate_graphs():
    # diagonal (n components), dense (a clique cannot be cut), empty
    n = 300
    d = _analyze(n, np.arange(n + 1), np.arange(n), np.ones(n), "nesdis")
    assert sorted(d["Perm"].tolist()) == list(range(n)) and d["lnz"] == n
    m = 150
    ii, jj = np.tril_indices(m)
    order = np.lexsort((ii, jj))
    Ap = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(jj[order], minlength=m), out=Ap[1:])
    Ax = np.where(ii[order] == jj[order], m + 1.0, 0.1)
    dd = _analyze(m, Ap, ii[order].astype(np.int64), Ax, "nesdis")
    assert sorted(dd["Perm"].tolist()) == list(range(m)) and dd["lnz"] == m * (m + 1) / 2
    e = _analyze(0, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0), "nesdis")
    assert e["Perm"].size == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["p3d_16", "rand600", "two_blocks"])
def test_factor_and_solve_on_the_default_ordering(name):
    n, Ap, Ai, Ax = CASES[name]()
    S = ch.Session(ordering="default")
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    O = OracleFactor(n, Ap, Ai, -1, perm=fv.Perm.copy(), postorder=True)
    assert O.factorize(Ax) == 0
    m = O.lower_mask()
    assert np.linalg.norm((fv.x - O.x)[m]) <= 1e-12 * np.linalg.norm(O.x[m])
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    r = G.sym_matvec(n, Ap, Ai, Ax, -1, x) - b
    assert np.linalg.norm(r) <= 1e-11 * np.linalg.norm(b)
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()
