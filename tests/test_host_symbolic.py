"""CPU tests of the product's host layer (no GPU): the library loads, exports
every declared symbol, and its symbolic analysis reproduces the oracle's
index maps bit for bit (the oracle being pinned to the reference by
tests/test_oracle_golden.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G


def test_library_exports_every_declared_symbol():
    L = ch.lib()
    for name in ch.API_SYMBOLS + ch.HIP_SYMBOLS:
        assert hasattr(L, name), name
    assert b"gfx950" in L.cholmod_hip_version()


def test_headers_and_symbol_lists_agree():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for h in ("cholmod.h", "cholmod_hip.h", "SuiteSparse_config.h"):
        txt = open(os.path.join(root, "include", h)).read()
        declared |= set(re.findall(r"\b(cholmod_(?:l|hip|gpu)_[a-z0-9_]+|SuiteSparse_(?:start|finish|malloc|calloc|realloc|free))\s*\(", txt))
    assert declared == set(ch.API_SYMBOLS + ch.HIP_SYMBOLS)


CASES = {
    "p2d_17": lambda: G.poisson2d(17) + (-1, None),
    "p3d_9_nd": lambda: G.poisson3d(9) + (-1, G.geometric_nd(9, 9, 9, 2)),
    "p3d_12_nd4": lambda: G.poisson3d(12) + (-1, G.geometric_nd(12, 12, 12, 4)),
    "p3d_10x7x5": lambda: G.poisson3d(10, 7, 5) + (-1, G.geometric_nd(10, 7, 5, 3)),
    "box6r2": lambda: G.box_stencil3d(6, 2) + (-1, G.geometric_nd(6, 6, 6, 2)),
    "p2d_40_nd": lambda: G.poisson2d(40) + (-1, G.geometric_nd(40, 40, 1, 4)),
}


def _maps_equal(fv, O):
    assert fv.nsuper == O.nsuper and fv.ssize == O.ssize and fv.xsize == O.xsize
    assert fv.maxcsize == O.maxcsize and fv.maxesize == O.maxesize
    for name in ("Perm", "ColCount", "super", "pi", "px", "s"):
        assert np.array_equal(getattr(fv, name), getattr(O, name)), name


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("postorder", [True, False])
def test_symbolic_maps_bit_exact_vs_oracle(case, postorder):
    n, Ap, Ai, Ax, stype, perm = CASES[case]()
    S = ch.Session(postorder=postorder, use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    fv = ch.FactorView(Lf)
    O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=postorder)
    _maps_equal(fv, O)
    assert S.cm.fl == O.fl and S.cm.lnz == O.lnz
    assert S.L.cholmod_l_check_factor(Lf, C.byref(S.cm)) == 1
    assert fv.is_super and fv.is_ll and fv.xtype == ch.PATTERN
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0 and S.cm.memory_inuse == 0
    S.finish()


def test_upper_stored_input_gives_same_maps(golden_dir):
    """cholmod_l_read_sparse returns symmetric files upper-stored (prefer_upper);
    analysis must not depend on which triangle is stored."""
    rec = json.load(open(os.path.join(golden_dir, "reference_recorded.json")))["bcsstk01"]
    S = ch.Session(use_gpu=0)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    fp = libc.fopen(os.path.join(golden_dir, "bcsstk01.tri").encode(), b"r")
    A = S.L.cholmod_l_read_sparse(fp, C.byref(S.cm))
    libc.fclose(fp)
    assert A and A.contents.stype == 1 and A.contents.nrow == 48
    assert S.L.cholmod_l_nnz(A, C.byref(S.cm)) == 224
    Lf = S.analyze(A, np.array(rec["Perm"]))
    fv = ch.FactorView(Lf)
    for key in ("nsuper", "ssize", "xsize", "maxcsize", "maxesize"):
        assert getattr(fv, key) == rec[key]
    assert np.array_equal(fv.super, rec["super"])
    assert np.array_equal(fv.pi, rec["pi"]) and np.array_equal(fv.px, rec["px"])
    assert np.array_equal(fv.s[:8], rec["s_head"])
    assert np.array_equal(fv.Perm, rec["Perm"])
    assert S.cm.fl == rec["fl"] and S.cm.lnz == rec["lnz"]
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_poisson40_nd_matches_reference_profile(golden_dir):
    rec = json.load(open(os.path.join(golden_dir, "reference_recorded.json")))["poisson3d_nd"]["40"]
    n, Ap, Ai, Ax = G.poisson3d(40)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, G.geometric_nd(40, 40, 40, 4))
    fv = ch.FactorView(Lf)
    assert fv.nsuper == rec["nsuper"]
    assert fv.super[-1] - fv.super[-2] == rec["root_cols"]
    O = OracleFactor(n, Ap, Ai, -1, perm=G.geometric_nd(40, 40, 40, 4), postorder=True)
    _maps_equal(fv, O)
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_host_only_plan_schedule_and_levels():
    """Engine scheduling logic without a device (CHOLMOD_HIP_PLAN_HOST_ONLY)."""
    n, Ap, Ai, Ax = G.poisson3d(12)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, G.geometric_nd(12, 12, 12, 4))
    fv = ch.FactorView(Lf)
    st = C.c_int(0)
    f = Lf.contents
    plan = S.L.cholmod_hip_plan_create(fv.n, fv.nsuper, f.super, f.pi, f.px, f.s,
                                       ch.HIP_PLAN_HOST_ONLY, C.byref(st))
    assert plan and st.value == 0
    sparent = np.empty(fv.nsuper, dtype=np.int64)
    level = np.empty(fv.nsuper, dtype=np.int64)
    assert S.L.cholmod_hip_get_maps(plan, sparent.ctypes.data, level.ctypes.data, None) == 0
    O = OracleFactor(n, Ap, Ai, -1, perm=G.geometric_nd(12, 12, 12, 4), postorder=True)
    assert np.array_equal(sparent, O.sparent())
    ok = sparent >= 0
    assert np.all(level[sparent[ok]] > level[ok])
    leaves = np.setdiff1d(np.arange(fv.nsuper), sparent[ok])
    assert np.all(level[leaves] == 0)
    stats = np.zeros(ch.CHOLMOD_HIP_NSTATS)
    assert S.L.cholmod_hip_get_stats(plan, stats.ctypes.data) == 0
    us = O.update_stats()
    assert abs(stats[1] / (us["update_flops"] + us["panel_flops"]) - 1) < 1e-12
    assert stats[5] == 8.0 * fv.xsize and stats[3] == level.max() + 1
    # factorizing without a device must fail loudly, not fall back
    minor = C.c_int64(0)
    assert S.L.cholmod_hip_factorize_resident(plan, 0.0, 0, C.byref(minor)) < 0
    S.L.cholmod_hip_plan_destroy(plan)
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_numeric_without_gpu_fails_loudly():
    """Common->useGPU = 1 without a device: loud failure by default; the CPU path
    only on request (Common->useGPU = 0, or the opted-in degradation)."""
    if ch.lib().cholmod_hip_probe():
        pytest.skip("a GPU is present")
    n, Ap, Ai, Ax = G.poisson2d(6)
    S = ch.Session(use_gpu=1)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A)
    assert S.factorize(A, Lf) == 0
    assert S.cm.status == ch.GPU_PROBLEM
    assert not Lf.contents.x                                # L is returned symbolic
    S.cm.hip_cpu_fallback = 1                               # the reference's degradation, opted in
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    assert np.linalg.norm(G.sym_matvec(n, Ap, Ai, Ax, -1, x) - b) < 1e-12 * np.linalg.norm(b)
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


def test_invalid_arguments():
    S = ch.Session(use_gpu=0)
    n, Ap, Ai, Ax = G.poisson2d(4)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    with pytest.raises(RuntimeError):
        S.analyze(A, np.zeros(n, dtype=np.int64))       # not a permutation
    assert S.cm.status == ch.INVALID
    # an unsymmetric matrix is analysed as A*A', with a column subset as A(:,f)*A(:,f)' (round 5, tests/test_aat.py);
    # a subset that lists a column twice is invalid
    A0 = S.sparse(n, Ap, Ai, Ax, 0)
    L0 = S.analyze(A0)
    assert L0 and S.cm.status == ch.OK and L0.contents.n == n
    S.free_factor(L0)
    fset = np.arange(3, dtype=np.int64)
    L1 = S.L.cholmod_l_analyze_p(A0, None, fset.ctypes.data, 3, C.byref(S.cm))
    assert L1 and S.cm.status == ch.OK and L1.contents.n == n
    S.free_factor(L1)
    fset = np.array([1, 2, 1], dtype=np.int64)
    assert not S.L.cholmod_l_analyze_p(A0, None, fset.ctypes.data, 3, C.byref(S.cm)) and S.cm.status == ch.INVALID
    S.free_sparse(A)
    S.free_sparse(A0)
    assert S.cm.malloc_count == 0
    S.finish()


def test_memory_budget_splits_the_sweep(monkeypatch):
    """With a tight arena budget the schedule sweeps subtrees one after the other:
    smaller arena, same executed flops, more launches."""
    n, Ap, Ai, Ax = G.poisson3d(24)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, G.geometric_nd(24, 24, 24, 4))
    fv = ch.FactorView(Lf)
    f = Lf.contents

    def plan_stats():
        st = C.c_int(0)
        plan = S.L.cholmod_hip_plan_create(fv.n, fv.nsuper, f.super, f.pi, f.px, f.s,
                                           ch.HIP_PLAN_HOST_ONLY, C.byref(st))
        assert plan and st.value == 0
        s24 = np.zeros(ch.CHOLMOD_HIP_NSTATS)
        S.L.cholmod_hip_get_stats(plan, s24.ctypes.data)
        S.L.cholmod_hip_plan_destroy(plan)
        return s24

    base = plan_stats()
    assert base[22] == 1
    monkeypatch.setenv("CHOLMOD_HIP_ARENA_BUDGET_MB", str(0.55 * base[4] / 1048576.0))
    tight = plan_stats()
    assert tight[22] > 1 and tight[4] < 0.75 * base[4]
    assert tight[1] == base[1] and tight[2] > base[2]
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


def test_for_whom_variants_of_super_symbolic():
    """cholmod_l_analyze_p2's for_whom (CHOLMOD/Supernodal/cholmod_super_symbolic.c:
    155-160, :662-663, :749-771, :912-913): FOR_SPQRGPU carries the same numeric
    sizes as FOR_CHOLESKY; plain FOR_SPQR only the row structure, px[0] = 123456."""
    n, Ap, Ai, Ax = G.poisson3d(9)
    perm = np.ascontiguousarray(G.geometric_nd(9, 9, 9, 3))
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    views = {}
    for who in (0, 1, 2):                       # FOR_SPQR, FOR_CHOLESKY, FOR_SPQRGPU
        Lf = S.L.cholmod_l_analyze_p2(who, A, perm.ctypes.data, None, 0, C.byref(S.cm))
        assert Lf and S.cm.status == ch.OK
        assert S.L.cholmod_l_check_factor(Lf, C.byref(S.cm)) == 1
        fv = ch.FactorView(Lf)
        views[who] = {k: np.array(getattr(fv, k)) for k in ("Perm", "super", "pi", "px", "s")}
        views[who].update(xsize=fv.xsize, maxcsize=fv.maxcsize, maxesize=fv.maxesize, useGPU=fv.useGPU)
        S.free_factor(Lf)
    for k in ("Perm", "super", "pi", "s"):
        assert np.array_equal(views[0][k], views[1][k]) and np.array_equal(views[2][k], views[1][k])
    assert np.array_equal(views[2]["px"], views[1]["px"])
    assert (views[2]["xsize"], views[2]["maxcsize"], views[2]["maxesize"]) == \
        (views[1]["xsize"], views[1]["maxcsize"], views[1]["maxesize"])
    assert views[0]["px"][0] == 123456 and views[0]["xsize"] == 1
    assert views[0]["maxcsize"] == 1 and views[0]["maxesize"] == 1
    assert views[0]["useGPU"] == 0 and views[2]["useGPU"] == 0
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()
