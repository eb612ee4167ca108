"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the
C-ABI library; the oracle is only the checker."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

pytestmark = pytest.mark.gpu

TOL_L = 1e-12          # ||L - L_ref||_F / ||L_ref||_F over the lower trapezoids (north_star)
TOL_RES = 1e-11        # ||Ax-b||_2 / ||b||_2


@pytest.fixture(scope="module")
def L():
    lib = ch.lib()
    assert lib.cholmod_hip_probe() == 1, "no HIP device visible"
    return lib


def rel_err_lower(x, xref, mask):
    return np.linalg.norm((x - xref)[mask]) / np.linalg.norm(xref[mask])


# ---- dense kernels in isolation -------------------------------------------------

@pytest.mark.parametrize("nsrow,nscol", [(7, 7), (40, 13), (64, 64), (65, 64), (200, 100),
                                         (333, 129), (700, 530), (900, 64), (1500, 1100), (2500, 1700)])
@pytest.mark.parametrize("flags", [0, 4, 32, 64, 128, 2048, 64 | 2048, 512, 512 | 128, 1024, 1024 | 128])   # tile128, no swizzle, 512- / 2048-wide outer blocks, zero-filled CBs, potrf launches of their own, no fused solve + update steps
def test_dense_partial_factorization(L, nsrow, nscol, flags):
    rng = np.random.default_rng(nsrow * 1000 + nscol)
    M = rng.standard_normal((nsrow, nsrow))
    Fm = M @ M.T + nsrow * np.eye(nsrow)
    F = np.asfortranarray(Fm.copy())
    info = C.c_int64(-1)
    rc = L.cholmod_hip_dense_partial_factor(F.ctypes.data, nsrow, nscol, flags, C.byref(info))
    assert rc == 0 and info.value == 0
    ref = np.linalg.cholesky(Fm[:nscol, :nscol])
    L11 = np.tril(F[:nscol, :nscol])
    assert np.linalg.norm(L11 - ref) / np.linalg.norm(ref) < 1e-13
    if nsrow > nscol:
        L21 = Fm[nscol:, :nscol] @ np.linalg.inv(ref).T
        assert np.linalg.norm(F[nscol:, :nscol] - L21) / np.linalg.norm(L21) < 1e-12
        Sc = Fm[nscol:, nscol:] - L21 @ L21.T
        got = np.tril(F[nscol:, nscol:])
        assert np.linalg.norm(got - np.tril(Sc)) / np.linalg.norm(np.tril(Sc)) < 1e-12
    # strictly upper part of the diagonal block is never written
    assert np.array_equal(np.triu(F[:nscol, :nscol], 1), np.triu(Fm[:nscol, :nscol], 1))


def test_dense_not_posdef_info(L):
    n = 150
    rng = np.random.default_rng(3)
    M = rng.standard_normal((n, n))
    Fm = M @ M.T + n * np.eye(n)
    Fm[97, 97] = -1.0
    F = np.asfortranarray(Fm.copy())
    info = C.c_int64(-1)
    assert L.cholmod_hip_dense_partial_factor(F.ctypes.data, n, n, 0, C.byref(info)) == 0
    assert info.value == 98                      # 1-based failing column, LAPACK convention
    ref = np.linalg.cholesky(Fm[:97, :97])
    assert np.linalg.norm(np.tril(F[:97, :97]) - ref) / np.linalg.norm(ref) < 1e-13
    ii, jj = np.indices((n, n))
    assert np.all(F[(jj >= 97) & (ii >= jj)] == 0)


# ---- sparse path -----------------------------------------------------------------

def _case(name, golden_dir):
    if name == "bcsstk01":
        rec = json.load(open(os.path.join(golden_dir, "reference_recorded.json")))["bcsstk01"]
        n, Ap, Ai, Ax, stype = G.read_triplet(os.path.join(golden_dir, "bcsstk01.tri"))
        return n, Ap, Ai, Ax, stype, np.array(rec["Perm"])
    if name == "bcsstk02":
        n, Ap, Ai, Ax, stype = G.read_triplet(os.path.join(golden_dir, "bcsstk02.tri"))
        return n, Ap, Ai, Ax, stype, None
    if name == "p2d_60_nd":
        return G.poisson2d(60) + (-1, G.geometric_nd(60, 60, 1, 4))
    if name == "p3d_12_nd":
        return G.poisson3d(12) + (-1, G.geometric_nd(12, 12, 12, 4))
    if name == "p3d_24_nd":
        return G.poisson3d(24) + (-1, G.geometric_nd(24, 24, 24, 4))
    if name == "p3d_11x9x14_nat":
        return G.poisson3d(11, 9, 14) + (-1, None)
    if name == "box9r2_nd":
        return G.box_stencil3d(9, 2) + (-1, G.geometric_nd(9, 9, 9, 3))
    if name == "p3d_32_nd":
        return G.poisson3d(32) + (-1, G.geometric_nd(32, 32, 32, 4))
    raise KeyError(name)


SPARSE_CASES = ["bcsstk01", "bcsstk02", "p2d_60_nd", "p3d_12_nd", "p3d_24_nd", "p3d_11x9x14_nat",
                "box9r2_nd", "p3d_32_nd"]


@pytest.mark.parametrize("name", SPARSE_CASES)
def test_factor_and_residual_match_oracle(L, golden_dir, name):
    n, Ap, Ai, Ax, stype, perm = _case(name, golden_dir)
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    assert O.factorize(Ax) == 0
    for k in ("Perm", "super", "pi", "px", "s"):
        assert np.array_equal(getattr(fv, k), getattr(O, k))
    mask = O.lower_mask()
    assert rel_err_lower(fv.x, O.x, mask) < TOL_L
    assert np.all(fv.x[~mask] == 0)              # dead upper triangles stay zero
    assert fv.minor == n and fv.xtype == ch.REAL
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    r = G.sym_matvec(n, Ap, Ai, Ax, stype, x) - b
    assert np.linalg.norm(r) / np.linalg.norm(b) < TOL_RES
    xo = O.solve(b)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-10
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


def test_bcsstk01_golden_values(L, golden_dir):
    rec = json.load(open(os.path.join(golden_dir, "reference_recorded.json")))["bcsstk01"]
    n, Ap, Ai, Ax, stype, perm = _case("bcsstk01", golden_dir)
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1
    fv = ch.FactorView(Lf)
    np.testing.assert_allclose(fv.x[:4], rec["Lx_head"], rtol=1e-13)
    np.testing.assert_allclose(np.linalg.norm(fv.x), rec["Lx_fro"], rtol=1e-13)      # (= sqrt (trace A): pins the input, not the reference)
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_relative_maps_bit_exact(L, golden_dir):
    for name in ("p3d_12_nd", "box9r2_nd", "bcsstk02"):
        n, Ap, Ai, Ax, stype, perm = _case(name, golden_dir)
        S = ch.Session()
        A = S.sparse(n, Ap, Ai, Ax, stype)
        Lf = S.analyze(A, perm)
        assert S.factorize(A, Lf) == 1
        fv = ch.FactorView(Lf)
        O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
        sparent = np.empty(fv.nsuper, dtype=np.int64)
        level = np.empty(fv.nsuper, dtype=np.int64)
        relmap = np.empty(max(fv.ssize - n, 1), dtype=np.int64)
        assert S.L.cholmod_hip_get_maps(fv.hip_plan, sparent.ctypes.data, level.ctypes.data,
                                        relmap.ctypes.data) == 0
        assert np.array_equal(sparent, O.sparent())
        ref = O.relmap_to_parent()
        # entries of parentless supernodes do not exist in either layout
        assert np.array_equal(relmap[:ref.size], ref)
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()


def test_plan_flag_variants_agree(L, golden_dir):
    n, Ap, Ai, Ax, stype, perm = _case("p3d_24_nd", golden_dir)
    xs = []
    for flags in (0, 4, 16, 32, 64, 128, 2048, 512, 1024, 4096):
        S = ch.Session(hip_flags=flags)
        A = S.sparse(n, Ap, Ai, Ax, stype)
        Lf = S.analyze(A, perm)
        assert S.factorize(A, Lf) == 1
        xs.append(ch.FactorView(Lf).x.copy())
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()
    for x in xs[1:]:
        assert np.linalg.norm(xs[0] - x) / np.linalg.norm(x) < 1e-13


@pytest.mark.parametrize("quick", [False, True])
def test_not_posdef_protocol_matches_oracle(L, quick):
    n, Ap, Ai, Ax = G.poisson3d(10)
    perm = G.geometric_nd(10, 10, 10, 3)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    sup = O.super
    # a failing pivot in the middle of a mid-tree supernode with several columns
    cand = [s for s in range(O.nsuper // 2, O.nsuper) if sup[s + 1] - sup[s] >= 6]
    sbad = cand[0]
    kbad = int(sup[sbad] + 3)
    Ax2 = Ax.copy()
    Ax2[Ap[int(O.Perm[kbad])]] = -7.0
    assert O.factorize(Ax2, quick_return=quick) == 1 and O.minor == kbad
    S = ch.Session()
    S.cm.quick_return_if_not_posdef = int(quick)
    A = S.sparse(n, Ap, Ai, Ax2, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1               # TRUE, as the reference
    assert S.cm.status == ch.NOT_POSDEF
    fv = ch.FactorView(Lf)
    assert fv.minor == kbad
    mask = O.lower_mask()
    nz = O.x != 0
    assert np.array_equal(fv.x[mask] != 0, nz[mask])
    assert rel_err_lower(fv.x, O.x, mask) < TOL_L
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_not_posdef_inside_a_leaf_front_on_the_mapped_path(L):
    """A pivot fails inside a leaf front during a refactorization of a resident S (new values,
    same pattern): the leaf level then runs two fronts to a wave (k_leaf_pair) -- minor, the
    zero pattern and the values against the oracle's repeat-supernode pass."""
    n, Ap, Ai, Ax = G.poisson2d(40)
    perm = G.geometric_nd(40, 40, 1, 4)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    sup, par = O.super, O.sparent()
    parents = set(int(p) for p in par if p >= 0)
    leaves = [s for s in range(O.nsuper) if s not in parents and sup[s + 1] - sup[s] >= 3]
    assert len(leaves) > 4
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK            # first factorization: records the map
    for sbad in (leaves[len(leaves) // 2], leaves[1]):
        kbad = int(sup[sbad] + 1)
        Ax2 = Ax.copy()
        Ax2[Ap[int(O.Perm[kbad])]] = -7.0
        assert O.factorize(Ax2) == 1 and O.minor == kbad
        ch._view(A.contents.x, len(Ax2), C.c_double, np.float64)[:] = Ax2
        assert S.factorize(A, Lf) == 1                                  # TRUE, as the reference
        assert S.cm.status == ch.NOT_POSDEF
        fv = ch.FactorView(Lf)
        assert fv.minor == kbad
        mask = O.lower_mask()
        assert np.array_equal(fv.x[mask] != 0, (O.x != 0)[mask])
        assert rel_err_lower(fv.x, O.x, mask) < TOL_L
    ch._view(A.contents.x, len(Ax), C.c_double, np.float64)[:] = Ax     # and positive definite again
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    assert O.factorize(Ax) == 0
    assert rel_err_lower(ch.FactorView(Lf).x, O.x, O.lower_mask()) < TOL_L
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_beta_refactorize_and_device_resident_mode(L, golden_dir):
    n, Ap, Ai, Ax, stype, perm = _case("p3d_12_nd", golden_dir)
    S = ch.Session(factor_on_device=True)
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf, beta=0.5) == 1
    assert not Lf.contents.x and Lf.contents.hip_on_device
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    r = G.sym_matvec(n, Ap, Ai, Ax, stype, x) + 0.5 * x - b
    assert np.linalg.norm(r) / np.linalg.norm(b) < TOL_RES
    assert S.refactorize_resident(Lf, beta=0.0) == 1
    assert S.L.cholmod_l_factor_to_host(Lf, C.byref(S.cm)) == 1
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    O.factorize(Ax)
    assert rel_err_lower(ch.FactorView(Lf).x, O.x, O.lower_mask()) < TOL_L
    st = S.hip_stats(Lf)
    assert st[0] > 0 and st[1] > 0
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_thin_fronts_through_the_assembly_map_with_beta(L):
    """2D grid (nearly every front is a thin one): the first factorization of a resident S
    searches the row lists and records the map, the later ones stream A through it
    (k_thin_front `mapped`); beta on the diagonal in both modes, each against the oracle."""
    n, Ap, Ai, Ax = G.poisson2d(48)
    perm = G.geometric_nd(48, 48, 1, 4)
    S = ch.Session(factor_on_device=True)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    mask = O.lower_mask()
    assert S.factorize(A, Lf, beta=0.125) == 1                  # search path, records the map
    for beta in (0.125, 0.25, 0.0):                             # mapped path
        if beta != 0.125:
            assert S.refactorize_resident(Lf, beta=beta) == 1
        assert S.L.cholmod_l_factor_to_host(Lf, C.byref(S.cm)) == 1
        assert O.factorize(Ax, beta=beta) == 0
        assert rel_err_lower(ch.FactorView(Lf).x, O.x, mask) < TOL_L, beta
        if beta == 0.125:
            assert S.refactorize_resident(Lf, beta=beta) == 1   # the same beta once more, now through the map
            assert S.L.cholmod_l_factor_to_host(Lf, C.byref(S.cm)) == 1
            assert rel_err_lower(ch.FactorView(Lf).x, O.x, mask) < TOL_L
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_triangular_solves_match_oracle(L, golden_dir):
    n, Ap, Ai, Ax, stype, perm = _case("p3d_12_nd", golden_dir)
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    O.factorize(Ax)
    rng = np.random.default_rng(0)
    b = rng.standard_normal((3, n))
    y = S.solve(Lf, b, ch.SYS_L)
    assert np.linalg.norm(y - O.lsolve(b)) / np.linalg.norm(y) < 1e-11
    z = S.solve(Lf, b, ch.SYS_Lt)
    assert np.linalg.norm(z - O.ltsolve(b)) / np.linalg.norm(z) < 1e-11
    p = S.solve(Lf, b[0], ch.SYS_P)
    assert np.array_equal(p, b[0][O.Perm])
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_update_kernel_microbench_runs(L):
    Pr = ch.probes()                # micro-benchmarks live in their own library
    for flags in (0, 4, 1):
        rate = Pr.cholmod_hip_bench_update_kernel(1024, 1024, 256, 2, flags)
        assert rate > 1e10
    assert Pr.cholmod_hip_bench_mfma_peak(2, 2000) > 1e13
    assert not hasattr(L, "cholmod_hip_bench_mfma_peak")      # ... and not in the product


def test_nothing_reads_the_arena_before_it_is_written(golden_dir, monkeypatch):
    """Contribution blocks are not zero-filled (the first trailing update of a front assigns them): with the arena
    poisoned with NaNs before every factorization (test hook), nothing of it may reach the factor -- whatever a
    fresh allocation happens to hold.  First factorization (search path) and one through the assembly map."""
    monkeypatch.setenv("CHOLMOD_HIP_TEST_POISON_ARENA", "1")
    for name in ("p3d_12_nd", "p2d_60_nd", "box9r2_nd"):
        n, Ap, Ai, Ax, stype, perm = _case(name, golden_dir)
        O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
        assert O.factorize(Ax) == 0
        S = ch.Session(hooks=True)      # (test hooks: lib/libcholmod_amd_testhooks.so)
        A = S.sparse(n, Ap, Ai, Ax, stype)
        Lf = S.analyze(A, perm)
        m = O.lower_mask()
        for _ in range(2):
            assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
            x = ch.FactorView(Lf).x
            assert np.all(np.isfinite(x)) and np.linalg.norm((x - O.x)[m]) / np.linalg.norm(O.x[m]) < 1e-12
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()


def test_wave_tile_kernel_16x16_super_tile_walk_is_bit_identical():
    """k_update3 walking 16 x 16 super-tiles (GemmGroup.swz == 2: what an XCD runs at a time with one wave per
    tile; the engine uses it for unshared regions of >= 8192 tiles) covers every tile exactly once: bitwise the
    result of k_update2's 8 x 8 walk, on ragged / triangular / assign regions of up to 61 x 79 tiles."""
    Pr = ch.probes()
    D4, SWZ16 = 32768, 262144
    for (m, n, k, tri, asg) in ((2049, 2049, 34, 1, 0), (4096, 4096, 16, 0, 0), (5000, 3000, 19, 1, 1), (3900, 3333, 9, 1, 0),
                                (64, 64, 8, 0, 0), (1100, 900, 12, 0, 1), (5056, 3900, 5, 0, 0)):
        assert Pr.cholmod_hip_debug_update_diff(m, n, k, tri, asg, D4 | SWZ16) == 0.0, (m, n, k, tri, asg)


def test_shim_level_factorize_with_host_copy(L, golden_dir):
    """The plain-pointer ABI exactly as INTEGRATION.md binds it: plan from the index
    maps, cholmod_hip_factorize with an Lx_host buffer, solve on the device."""
    n, Ap, Ai, Ax, stype, perm = _case("p3d_12_nd", golden_dir)
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    fv = ch.FactorView(Lf)
    f = Lf.contents
    # S = tril(P A P') by the library's own permuted transposes
    A1 = S.L.cholmod_l_ptranspose(A, 2, None, None, 0, C.byref(S.cm))
    Sm = S.L.cholmod_l_ptranspose(A1, 2, fv.Perm.ctypes.data, None, 0, C.byref(S.cm))
    st = C.c_int(0)
    plan = L.cholmod_hip_plan_create(fv.n, fv.nsuper, f.super, f.pi, f.px, f.s, 0, C.byref(st))
    assert plan and st.value == 0
    Lx = np.full(fv.xsize, np.nan)
    minor = C.c_int64(-1)
    sm = Sm.contents
    rc = L.cholmod_hip_factorize(plan, sm.p, sm.i, None, sm.x, 0.0, 0, Lx.ctypes.data, C.byref(minor))
    assert rc == 0 and minor.value == n
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    O.factorize(Ax)
    assert rel_err_lower(Lx, O.x, O.lower_mask()) < TOL_L
    y = np.ascontiguousarray(G.demo_rhs(n)[O.Perm])
    assert L.cholmod_hip_solve(plan, 0, y.ctypes.data, 1, n) == 0
    x = np.empty(n); x[O.Perm] = y
    r = G.sym_matvec(n, Ap, Ai, Ax, stype, x) - G.demo_rhs(n)
    assert np.linalg.norm(r) / np.linalg.norm(G.demo_rhs(n)) < TOL_RES
    stats = np.zeros(ch.CHOLMOD_HIP_NSTATS)
    assert L.cholmod_hip_get_stats(plan, stats.ctypes.data) == 0 and stats[0] > 0
    L.cholmod_hip_plan_destroy(plan)
    S.free_sparse(Sm); S.free_sparse(A1)
    S.free_factor(Lf); S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


def test_subtree_sweep_schedule_matches_oracle(L, golden_dir, monkeypatch):
    """Force the memory-aware schedule (subtrees swept one after the other) and
    check the factor is unchanged."""
    n, Ap, Ai, Ax, stype, perm = _case("p3d_24_nd", golden_dir)
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
    O.factorize(Ax)
    monkeypatch.setenv("CHOLMOD_HIP_ARENA_BUDGET_MB", "8")
    S = ch.Session()
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    assert S.hip_stats(Lf)[22] > 1
    assert rel_err_lower(ch.FactorView(Lf).x, O.x, O.lower_mask()) < TOL_L
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    assert np.linalg.norm(G.sym_matvec(n, Ap, Ai, Ax, stype, x) - b) / np.linalg.norm(b) < TOL_RES
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()
