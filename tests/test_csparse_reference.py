"""The oracle and the product against REFERENCE CODE COMPILED HERE: the reference tree's CSparse (oracle/Makefile `ref`,
oracle/csref.py) -- its elimination tree, column counts, symbolic and up-looking numeric Cholesky (cs_etree.c, cs_post.c,
cs_counts.c, cs_schol.c, cs_chol.c).  CHOLMOD's own sources cannot be built in this image; CSparse is
another algorithm of the same reference for the same factor of P A P', so for every case below, under the factor's final
permutation,

  * Parent (the etree) and ColCount are IDENTICAL to cs_etree / cs_counts,
  * every entry of cs_chol's pattern of L lies inside the supernodal structure (super, pi, s), and with
    nrelax = zrelax = 0 (fundamental supernodes) the two patterns are the same set,
  * the values agree entry for entry (1e-12 of ||L||_F; measured 1e-16 .. 1e-15), the explicit zeros that relaxed
    amalgamation stores are exact zeros,
  * cholmod_l_solve returns what the reference's cs_cholsol steps (cs_ipvec, cs_lsolve, cs_ltsolve, cs_pvec) return.

The library is prebuilt (it is never built from anything but /root/reference/CSparse and travels to the GPU box as a file);
without it the module is skipped."""
import os

import numpy as np
import pytest

from oracle import csref
from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

pytestmark = pytest.mark.skipif(not csref.available(), reason="oracle/_ref/libcsparse_ref.so not built (make -C oracle ref)")

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-12


def _ldl(name):
    from ldl_files import read_ldl_matrix, upper_csc
    m = read_ldl_matrix(os.path.join(HERE, "golden", "ldl", name))
    Ap, Ai, Ax = upper_csc(m)
    return m["n"], Ap, Ai, Ax, 1, m["P"]


def _golden(name):
    n, Ap, Ai, Ax, stype = G.read_triplet(os.path.join(HERE, "golden", name))
    return n, Ap, Ai, Ax, stype, None


def _spd_file(d, f):
    import matrix_files as MF
    n, Lp, Li, Lx = MF.to_lower(MF.read_file(os.path.join(HERE, "golden", d, f)))
    return n, Lp, Li, np.asarray(Lx, dtype=np.float64), -1, None


def _grid(kind, m, leaf):
    if kind == "p3d":
        n, Ap, Ai, Ax = G.poisson3d(m)
        return n, Ap, Ai, Ax, -1, G.geometric_nd(m, m, m, leaf)
    if kind == "p2d":
        n, Ap, Ai, Ax = G.poisson2d(m)
        return n, Ap, Ai, Ax, -1, G.geometric_nd(m, m, 1, leaf)
    n, Ap, Ai, Ax = G.box_stencil3d(m, 2)
    return n, Ap, Ai, Ax, -1, G.geometric_nd(m, m, m, leaf)


CASES = {
    "bcsstk01": lambda: _golden("bcsstk01.tri"),
    "bcsstk02": lambda: _golden("bcsstk02.tri"),
    "ldl_A13_bcsstk01_stored_perm": lambda: _ldl("A13"),
    "ldl_A14_bcsstk01_jumbled": lambda: _ldl("A14"),
    "ldl_A17_mesh1e1": lambda: _ldl("A17"),
    "ldl_A22_bcsstk02_jumbled": lambda: _ldl("A22"),
    "poisson3d_12_nd": lambda: _grid("p3d", 12, 3),
    "poisson3d_9_natural": lambda: _grid("p3d", 9, 3)[:5] + (None,),
    "poisson2d_40_nd": lambda: _grid("p2d", 40, 4),
    "box_stencil_8_nd": lambda: _grid("box", 8, 3),
}
VARIANTS = [("default", True), ("default", False), ("norelax", True), ("norelax", False)]


def scatter_reference(R, sup, pi, px, s, xsize):
    """cs_chol's L (CSC of P A P') laid out as the supernodal factor: (values, in_pattern, entries that found no place)"""
    ref = np.zeros(xsize)
    inpat = np.zeros(xsize, dtype=bool)
    stray = 0
    for k in range(len(sup) - 1):
        k1, k2 = int(sup[k]), int(sup[k + 1])
        rows = s[pi[k]:pi[k + 1]]
        nsrow = len(rows)
        for j in range(k1, k2):
            a, b = int(R["Lp"][j]), int(R["Lp"][j + 1])
            li = R["Li"][a:b]
            q = np.searchsorted(rows, li)
            ok = (q < nsrow) & (rows[np.minimum(q, nsrow - 1)] == li)
            stray += int((~ok).sum())
            off = int(px[k]) + (j - k1) * nsrow + q[ok]
            ref[off] = R["Lx"][a:b][ok]
            inpat[off] = True
    return ref, inpat, stray


def lower_mask(sup, pi, px, xsize):
    m = np.zeros(xsize, dtype=bool)
    for k in range(len(sup) - 1):
        nscol, nsrow = int(sup[k + 1] - sup[k]), int(pi[k + 1] - pi[k])
        blk = np.arange(nsrow)[:, None] >= np.arange(nscol)[None, :]
        m[int(px[k]):int(px[k]) + nsrow * nscol] = blk.T.ravel()
    return m


def check_against_reference(case, Perm, Parent, ColCount, sup, pi, px, s, x, relax, what):
    n, Ap, Ai, Ax, stype, _ = case
    R = csref.reference_cholesky(n, Ap, Ai, Ax, stype, perm=Perm)
    assert R["ok"], "the reference finds the matrix not positive definite"
    if Parent is not None:
        assert np.array_equal(np.where(np.asarray(Parent) < 0, -1, Parent), R["parent"]), what + ": etree"
    assert np.array_equal(ColCount, R["colcount"]), what + ": column counts"
    ref, inpat, stray = scatter_reference(R, sup, pi, px, s, len(x))
    assert stray == 0, what + ": an entry of the reference's L has no place in the supernodal structure"
    m = lower_mask(sup, pi, px, len(x))
    assert int(inpat.sum()) == R["lnz"] and not np.any(inpat & ~m)
    if relax == "norelax":
        assert np.array_equal(inpat, m), what + ": fundamental supernodes store exactly the pattern of L"
    assert np.linalg.norm((x - ref)[m]) <= TOL * np.linalg.norm(ref[m]), what + ": values of L"
    assert not np.any(x[m & ~inpat]), what + ": explicit zeros of the amalgamation"
    assert not np.any(x[~m]), what + ": above the diagonal blocks"
    return R


@pytest.mark.parametrize("relax,postorder", VARIANTS)
@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_the_compiled_reference(name, relax, postorder):
    case = CASES[name]()
    n, Ap, Ai, Ax, stype, perm = case
    kw = {"nrelax": [0, 0, 0], "zrelax": [0.0, 0.0, 0.0]} if relax == "norelax" else {}
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=postorder, **kw)
    assert O.factorize(Ax) == 0
    x = O.x.copy()
    x[~O.lower_mask()] = 0.0        # (what LAPACK leaves above a diagonal block is not part of L)
    check_against_reference(case, O.Perm, O.Parent, O.ColCount, O.super, O.pi, O.px, O.s, x, relax, "oracle")


def _product(case, relax, postorder, use_gpu):
    n, Ap, Ai, Ax, stype, perm = case
    S = ch.Session(use_gpu=use_gpu, postorder=postorder)
    if relax == "norelax":
        for k in range(3):
            S.cm.nrelax[k] = 0
            S.cm.zrelax[k] = 0.0
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    what = "HIP path" if use_gpu else "CPU path"
    R = check_against_reference(case, fv.Perm, None, fv.ColCount, fv.super, fv.pi, fv.px, fv.s, fv.x, relax, what)
    assert int(S.cm.lnz) == R["lnz"]
    # the solve (cholmod_l_solve: super_lsolve / ltsolve under the permutation) against the reference's cs_cholsol steps
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, 3))
    Xr = csref.reference_cholesky(n, Ap, Ai, Ax, stype, perm=fv.Perm, b=B)["x"]
    for k in range(3):
        x = S.solve(Lf, B[:, k].copy())
        assert np.linalg.norm(x - Xr[:, k]) <= 1e-9 * np.linalg.norm(Xr[:, k]), what + ": solve"
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


@pytest.mark.parametrize("relax,postorder", VARIANTS)
@pytest.mark.parametrize("name", sorted(CASES))
def test_cpu_path_against_the_compiled_reference(name, relax, postorder):
    _product(CASES[name](), relax, postorder, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("relax,postorder", VARIANTS)
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_path_against_the_compiled_reference(name, relax, postorder):
    _product(CASES[name](), relax, postorder, 1)


# ---- the reference's own positive definite test matrices (tests/golden/tcov, tests/golden/demo) ---------------------------
def _spd_files():
    import matrix_files as MF
    out = []
    for d in ("tcov", "demo"):
        dd = os.path.join(HERE, "golden", d)
        for f in sorted(os.listdir(dd)):
            try:
                m = MF.read_file(os.path.join(dd, f))
            except Exception:           # noqa: BLE001  (not every file is a matrix the readers take: covered in test_reader_files)
                continue
            if (m.get("kind") != "sparse" or m["xtype"] != "real" or m["nrow"] != m["ncol"] or m["stype"] == 0
                    or not 0 < m["nrow"] <= 2000):
                continue
            out.append((d, f))
    return out


@pytest.mark.parametrize("d,f", _spd_files())
def test_reference_matrices_against_the_compiled_reference(d, f):
    """every real symmetric file of the reference's Tcov / Demo directories that IS positive definite (by the reference's own
    cs_chol): oracle and CPU path against it; the indefinite ones must be refused by both (status NOT_POSDEF)"""
    case = _spd_file(d, f)
    n, Ap, Ai, Ax, stype, perm = case
    if not np.all(np.isfinite(Ax)):
        pytest.skip("non-finite entries")
    R = csref.reference_cholesky(n, Ap, Ai, Ax, stype)
    O = OracleFactor(n, Ap, Ai, stype, perm=None, postorder=True)
    info = O.factorize(Ax)
    if not R["ok"]:
        assert info != 0
        return
    if info != 0:
        # positive definite for the up-looking reference, a pivot <= 0 in the supernodal order of summation: only at the
        # edge of definiteness
        d_ = np.sort(np.abs(R["Lx"][R["Lp"][:-1]]))
        assert (d_[0] / d_[-1]) ** 2 < 1e-12
        return
    x = O.x.copy()
    x[~O.lower_mask()] = 0.0
    dg = np.abs(R["Lx"][R["Lp"][:-1]])
    if (dg.min() / dg.max()) ** 2 < 1e-10:
        pytest.skip("ill-conditioned: entrywise agreement of two orders of summation is not defined")
    check_against_reference(case, O.Perm, O.Parent, O.ColCount, O.super, O.pi, O.px, O.s, x, "default", "oracle")
    _product(case, "default", True, 0)


# ---- a grid of the size of the engine's mid-size fronts (n = 32 768, 8.2e6 entries of L, fronts of up to ~1 600 rows) ---------
def test_oracle_on_poisson_32_cubed_against_the_compiled_reference():
    case = _grid("p3d", 32, 4)
    n, Ap, Ai, Ax, stype, perm = case
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    assert O.factorize(Ax) == 0
    x = O.x.copy()
    x[~O.lower_mask()] = 0.0
    R = check_against_reference(case, O.Perm, O.Parent, O.ColCount, O.super, O.pi, O.px, O.s, x, "default", "oracle")
    assert R["lnz"] == int(O.lnz) == 8245617


@pytest.mark.gpu
def test_hip_path_on_poisson_32_cubed_against_the_compiled_reference():
    _product(_grid("p3d", 32, 4), "default", True, 1)


# ---- seeded random matrices: patterns no grid has (dense rows, isolated vertices, chains), random permutations ---------------
def _random_spd(seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 260))
    density = float(rng.choice([0.01, 0.03, 0.08, 0.2]))
    Bm = sp.random(n, n, density=density, random_state=np.random.RandomState(seed), format="csc")
    if rng.random() < 0.3:                                 # a dense row / column
        k = int(rng.integers(0, n))
        Bm = Bm.tolil()
        Bm[k, :] = rng.standard_normal(n) * (rng.random(n) < 0.6)
        Bm = Bm.tocsc()
    Am = (Bm + Bm.T).tocsc()
    Am.setdiag(0)
    Am.eliminate_zeros()
    rowsum = np.asarray(abs(Am).sum(axis=1)).ravel()
    Am = (Am + sp.diags(rowsum + rng.random(n) + 0.1)).tocsc()          # strictly diagonally dominant: positive definite
    L = sp.tril(Am, format="csc")
    L.sort_indices()
    perm = rng.permutation(n).astype(np.int64) if rng.random() < 0.7 else None
    return n, L.indptr.astype(np.int64), L.indices.astype(np.int64), L.data.astype(np.float64), -1, perm


@pytest.mark.parametrize("seed", range(40))
def test_random_matrices_oracle_and_cpu_path_against_the_compiled_reference(seed):
    case = _random_spd(1000 + seed)
    n, Ap, Ai, Ax, stype, perm = case
    relax = "norelax" if seed % 3 == 0 else "default"
    postorder = seed % 2 == 0
    kw = {"nrelax": [0, 0, 0], "zrelax": [0.0, 0.0, 0.0]} if relax == "norelax" else {}
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=postorder, **kw)
    assert O.factorize(Ax) == 0
    x = O.x.copy()
    x[~O.lower_mask()] = 0.0
    check_against_reference(case, O.Perm, O.Parent, O.ColCount, O.super, O.pi, O.px, O.s, x, relax, "oracle")
    _product(case, relax, postorder, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_random_matrices_hip_path_against_the_compiled_reference(seed):
    _product(_random_spd(1000 + seed), "norelax" if seed % 3 == 0 else "default", seed % 2 == 0, 1)


@pytest.mark.gpu
@pytest.mark.slow
def test_hip_path_on_poisson_40_cubed_against_the_compiled_reference():
    """(-m "gpu and slow": 32 s of cs_chol + the entry-by-entry comparison of 2.2e7 entries)"""
    _product(_grid("p3d", 40, 4), "default", True, 1)
