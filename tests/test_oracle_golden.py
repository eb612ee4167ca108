"""Pins the CPU oracle (oracle/ssoracle.c) against the reference outputs recorded
in SURVEY.md 8c/8d (tests/golden/reference_recorded.json) and against the
structural invariants of cholmod_l_check_factor
(reference CHOLMOD/Check/cholmod_check.c:1823-2000)."""
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleFactor
from suitesparse_amd import generators as G


@pytest.fixture(scope="module")
def rec(golden_dir):
    with open(os.path.join(golden_dir, "reference_recorded.json")) as f:
        return json.load(f)


def check_factor_invariants(L):
    sup, pi, px, s = L.super, L.pi, L.px, L.s
    assert pi[0] == 0 and max(1, pi[-1]) == L.ssize
    assert px[0] == 0 and max(1, px[-1]) == L.xsize
    for k in range(L.nsuper):
        k1, k2 = sup[k], sup[k + 1]
        nscol, nsrow = k2 - k1, pi[k + 1] - pi[k]
        assert 0 <= k1 < k2 <= L.n and nsrow >= nscol
        assert px[k + 1] - px[k] == nsrow * nscol
        rows = s[pi[k]:pi[k + 1]]
        assert np.array_equal(rows[:nscol], np.arange(k1, k2))
        assert np.all(np.diff(rows) > 0) and rows[-1] < L.n
    assert np.array_equal(np.sort(L.Perm), np.arange(L.n))
    cc = L.ColCount
    assert np.all(cc >= 0) and np.all(cc <= L.n - np.arange(L.n))


@pytest.mark.parametrize("postorder", [False, True])
def test_bcsstk01_matches_reference(rec, golden_dir, postorder):
    g = rec["bcsstk01"]
    n, Ap, Ai, Ax, stype = G.read_triplet(os.path.join(golden_dir, g["matrix"]))
    assert (n, int(Ap[-1]), stype) == (48, 224, -1)
    L = OracleFactor(n, Ap, Ai, stype, perm=np.array(g["Perm"]), postorder=postorder)
    # GIVEN final Perm reproduces the maps with postorder on or off
    assert np.array_equal(L.Perm, g["Perm"])
    for key in ("nsuper", "ssize", "xsize", "maxcsize", "maxesize"):
        assert getattr(L, key) == g[key], key
    assert np.array_equal(L.super, g["super"])
    assert np.array_equal(L.pi, g["pi"])
    assert np.array_equal(L.px, g["px"])
    assert np.array_equal(L.s[:8], g["s_head"])
    assert L.fl == g["fl"] and L.lnz == g["lnz"]
    check_factor_invariants(L)
    assert L.factorize(Ax) == 0 and L.minor == n
    np.testing.assert_allclose(L.x[:4], g["Lx_head"], rtol=1e-14)
    # (NOT a pin of the reference: ||L||_F^2 = trace (A) for any Cholesky factor under any permutation -- this checks
    # the input file and the identity, nothing about the algorithm)
    np.testing.assert_allclose(np.linalg.norm(L.x), g["Lx_fro"], rtol=1e-14)
    c = g["blas_calls"]
    assert list(L.calls) == [c["syrk"], c["gemm"], c["potrf"], c["trsm"]]
    b = G.demo_rhs(n)
    x = L.solve(b)
    r = G.sym_matvec(n, Ap, Ai, Ax, stype, x) - b
    assert np.linalg.norm(r) / np.linalg.norm(b) < g["resid_2norm_max"]


def test_poisson40_nd_profile_matches_reference(rec):
    g = rec["poisson3d_nd"]["40"]
    n, Ap, Ai, Ax = G.poisson3d(40)
    L = OracleFactor(n, Ap, Ai, -1, perm=G.geometric_nd(40, 40, 40, 4), postorder=True)
    st = L.update_stats()
    assert L.nsuper == g["nsuper"]
    assert int(st["updates"]) == g["updates"]
    assert L.super[-1] - L.super[-2] == g["root_cols"]
    assert abs(L.fl / g["fl"] - 1) < 5e-3 and abs(L.lnz / g["lnz"] - 1) < 5e-3
    assert abs((st["update_flops"] + st["panel_flops"]) / g["exec_flops"] - 1) < 5e-3
    assert abs(L.maxcsize / g["maxcsize"] - 1) < 2e-2
    assert abs(L.xsize * 8 / 1e9 / g["xsize_GB"] - 1) < 2e-2
    check_factor_invariants(L)


@pytest.mark.slow
def test_poisson100_nd_profile_matches_reference(rec):
    g = rec["poisson3d_nd"]["100"]
    n, Ap, Ai, Ax = G.poisson3d(100)
    L = OracleFactor(n, Ap, Ai, -1, perm=G.geometric_nd(100, 100, 100, 4), postorder=True)
    st = L.update_stats()
    assert L.nsuper == g["nsuper"]
    assert int(st["updates"]) == g["updates"]
    assert L.super[-1] - L.super[-2] == g["root_cols"]
    assert abs(L.fl / g["fl"] - 1) < 5e-3 and abs(L.xsize / g["xsize"] - 1) < 5e-3
    assert abs(L.maxcsize / g["maxcsize"] - 1) < 5e-3


@pytest.mark.parametrize("case", ["p2d_9", "p3d_6", "p3d_7_nd", "box5r2", "bcsstk02"])
def test_factor_is_cholesky_of_permuted_matrix(case, golden_dir):
    """L L' == P A P' densely (mathematical pin, independent of any reference)."""
    if case == "p2d_9":
        n, Ap, Ai, Ax = G.poisson2d(9); perm = None; stype = -1
    elif case == "p3d_6":
        n, Ap, Ai, Ax = G.poisson3d(6); perm = None; stype = -1
    elif case == "p3d_7_nd":
        n, Ap, Ai, Ax = G.poisson3d(7); perm = G.geometric_nd(7, 7, 7, 2); stype = -1
    elif case == "box5r2":
        n, Ap, Ai, Ax = G.box_stencil3d(5, 2); perm = G.geometric_nd(5, 5, 5, 2); stype = -1
    else:
        n, Ap, Ai, Ax, stype = G.read_triplet(os.path.join(golden_dir, "bcsstk02.tri")); perm = None
    L = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    check_factor_invariants(L)
    assert L.factorize(Ax) == 0
    A = np.zeros((n, n))
    cols = np.repeat(np.arange(n), np.diff(Ap))
    A[Ai, cols] = Ax
    A = A + A.T - np.diag(np.diag(A))
    P = L.Perm
    PAP = A[np.ix_(P, P)]
    Ld = np.zeros((n, n))
    sup, pi, px, s, x = L.super, L.pi, L.px, L.s, L.x
    for k in range(L.nsuper):
        nscol, nsrow = sup[k + 1] - sup[k], pi[k + 1] - pi[k]
        blk = x[px[k]:px[k] + nsrow * nscol].reshape(nscol, nsrow).T
        rows = s[pi[k]:pi[k + 1]]
        Ld[rows[:, None], np.arange(sup[k], sup[k + 1])[None, :]] = blk
    Ld = np.tril(Ld)
    err = np.linalg.norm(Ld @ Ld.T - PAP) / np.linalg.norm(PAP)
    assert err < 1e-13
    ref = np.linalg.cholesky(PAP)
    assert np.linalg.norm(Ld - ref) / np.linalg.norm(ref) < 1e-12


def test_not_posdef_protocol():
    """reference t_cholmod_super_numeric.c:905-968: status NOT_POSDEF, L->minor,
    leading columns of the failing supernode kept, everything after zero."""
    n, Ap, Ai, Ax = G.poisson2d(6)
    L = OracleFactor(n, Ap, Ai, -1, perm=None, postorder=True)
    Ax = Ax.copy()
    # make pivot of original column Perm[k*] negative: pick a column in the
    # middle of the last supernode so that a partial refactorization happens
    sup = L.super
    last = L.nsuper - 1
    kbad = int(sup[last] + (sup[last + 1] - sup[last]) // 2)
    orig = int(L.Perm[kbad])
    Ax[Ap[orig]] = -50.0        # diagonal is first in each lower-stored column
    st = L.factorize(Ax)
    assert st == 1 and L.minor == kbad
    px, pi = L.px, L.pi
    nsrow = pi[last + 1] - pi[last]
    blk = L.x[px[last]:px[last + 1]].reshape(-1, nsrow)
    ngood = kbad - sup[last]
    assert np.all(blk[ngood:] == 0)
    assert np.all(np.diag(blk[:ngood, :ngood]) > 0)
    # quick return leaves the whole failing supernode zero
    L2 = OracleFactor(n, Ap, Ai, -1, perm=None, postorder=True)
    assert L2.factorize(Ax, quick_return=True) == 1
    assert np.all(L2.x[px[last]:] == 0)


def test_relmap_and_sparent():
    n, Ap, Ai, Ax = G.poisson3d(6)
    L = OracleFactor(n, Ap, Ai, -1, perm=G.geometric_nd(6, 6, 6, 2), postorder=True)
    sp = L.sparent()
    rm = L.relmap_to_parent()
    sup, pi, s = L.super, L.pi, L.s
    assert np.all((sp == -1) | (sp > np.arange(L.nsuper)))
    for d in range(L.nsuper):
        nscol = sup[d + 1] - sup[d]
        rows = s[pi[d] + nscol:pi[d + 1]]
        if sp[d] < 0:
            assert rows.size == 0
            continue
        prow = s[pi[sp[d]]:pi[sp[d] + 1]]
        off = pi[d] - sup[d]
        assert np.array_equal(prow[rm[off:off + rows.size]], rows)
