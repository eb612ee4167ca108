"""Complex / zomplex input of the hot path (reference
CHOLMOD/Supernodal/t_cholmod_super_numeric.c:41-83, the COMPLEX and ZOMPLEX templates:
"A and F are complex or zomplex, L and C are complex").

The product computes the complex factor through the real embedding
(suitesparse_amd/csrc/host/complex.c); the oracle restates the reference's complex
template with zherk / zgemm / zpotrf / ztrsm (oracle/ssoracle.c, orc_factorize_complex).
Bars: index maps bit-exact, ||L - L_ref|| / ||L_ref|| < 1e-12 over the lower
trapezoids, dead upper triangles exactly zero, L L^H = P A P^H densely on the small
cases, residual < 1e-11, identical not-positive-definite state.

The CPU cases (Common->useGPU == 0) run without a GPU; the `gpu` cases run the same
checks through the HIP engine."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

TOL_L = 1e-12
TOL_RES = 1e-11


def hermitian_from(n, Ap, Ai, Ax, stype=-1, seed=0, scale=0.9):
    """A Hermitian positive definite matrix on the pattern of a real SPD one: every
    off-diagonal entry is turned by a random phase and shrunk (the diagonal keeps
    dominating).  Returns the lower-stored CSC (Ap, Ai, complex Ax), sorted."""
    rng = np.random.default_rng(seed)
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    Al = (sp.tril(A) if stype < 0 else sp.triu(A).T).tocsc()
    Al.sort_indices()
    vals = Al.data.astype(np.complex128)
    cols = np.repeat(np.arange(n), np.diff(Al.indptr))
    off = Al.indices != cols
    vals[off] *= scale * np.exp(1j * rng.uniform(0, 2 * np.pi, int(off.sum())))
    return Al.indptr.astype(np.int64), Al.indices.astype(np.int64), vals


def full_hermitian(n, Ap, Ai, Ax):
    Al = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    return (Al + sp.tril(Al, -1).conj().T).tocsr()


def dense_L(fv_or_oracle, x):
    n = fv_or_oracle.n
    Ld = np.zeros((n, n), dtype=x.dtype)
    sup, pi, px, s = (np.asarray(getattr(fv_or_oracle, k)) for k in ("super", "pi", "px", "s"))
    for k in range(len(sup) - 1):
        nscol, nsrow = int(sup[k + 1] - sup[k]), int(pi[k + 1] - pi[k])
        blk = x[px[k]:px[k] + nsrow * nscol].reshape(nscol, nsrow).T
        r = s[pi[k]:pi[k + 1]]
        for j in range(nscol):
            Ld[r[j:], sup[k] + j] = blk[j:, j]
    return Ld


def _case(name, golden_dir):
    if name == "bcsstk01":
        rec = json.load(open(os.path.join(golden_dir, "reference_recorded.json")))["bcsstk01"]
        n, Ap, Ai, Ax, stype = G.read_triplet(os.path.join(golden_dir, "bcsstk01.tri"))
        return (n,) + hermitian_from(n, Ap, Ai, Ax, stype, scale=0.5) + (np.array(rec["Perm"]),)
    if name == "p3d_9_nd":
        n, Ap, Ai, Ax = G.poisson3d(9)
        return (n,) + hermitian_from(n, Ap, Ai, Ax) + (G.geometric_nd(9, 9, 9, 3),)
    if name == "box7r2_nd":
        n, Ap, Ai, Ax = G.box_stencil3d(7, 2)
        return (n,) + hermitian_from(n, Ap, Ai, Ax, scale=0.3) + (G.geometric_nd(7, 7, 7, 3),)
    if name == "p2d_40_nat":
        n, Ap, Ai, Ax = G.poisson2d(40)
        return (n,) + hermitian_from(n, Ap, Ai, Ax) + (None,)
    if name == "p3d_24_nd":
        n, Ap, Ai, Ax = G.poisson3d(24)
        return (n,) + hermitian_from(n, Ap, Ai, Ax) + (G.geometric_nd(24, 24, 24, 4),)
    if name == "box16r2_nd":
        n, Ap, Ai, Ax = G.box_stencil3d(16, 2)
        return (n,) + hermitian_from(n, Ap, Ai, Ax, scale=0.3) + (G.geometric_nd(16, 16, 16, 4, 2),)
    raise KeyError(name)


def _check(name, golden_dir, use_gpu, zomplex=False, upper=False, dense_check=True, session_kwargs=None):
    n, Ap, Ai, Ax, perm = _case(name, golden_dir)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    assert O.factorize_complex(Ax, zomplex=zomplex) == 0
    S = ch.Session(use_gpu=use_gpu, **(session_kwargs or {}))
    if upper:
        # the same Hermitian matrix with its upper triangle stored: exercises the
        # conjugate permuted transpose of cholmod_l_factorize (cholmod_factorize.c:225-232)
        Au = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)).conj().T.tocsc()
        Au.sort_indices()
        A = S.sparse(n, Au.indptr.astype(np.int64), Au.indices.astype(np.int64), Au.data, 1, zomplex=zomplex)
    else:
        A = S.sparse(n, Ap, Ai, Ax, -1, zomplex=zomplex)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    fv = ch.FactorView(Lf)
    assert fv.xtype == ch.COMPLEX and fv.minor == n
    for k in ("Perm", "ColCount", "super", "pi", "px", "s"):
        assert np.array_equal(getattr(fv, k), getattr(O, k)), k
    mask = O.lower_mask()
    err = np.linalg.norm((fv.x - O.xc)[mask]) / np.linalg.norm(O.xc[mask])
    assert err < TOL_L, err
    assert np.all(fv.x[~mask] == 0)                  # dead upper triangles stay zero
    assert S.L.cholmod_l_check_factor(Lf, C.byref(S.cm)) == 1
    Af = full_hermitian(n, Ap, Ai, Ax)
    if dense_check and n <= 2000:
        Ld = dense_L(fv, fv.x)
        P = fv.Perm
        Ad = Af.toarray()[np.ix_(P, P)]
        assert np.linalg.norm(Ld @ Ld.conj().T - Ad) / np.linalg.norm(Ad) < 1e-13
        assert np.all(np.diag(Ld).imag == 0) and np.all(np.diag(Ld).real > 0)
    rng = np.random.default_rng(7)
    b = rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))
    x = S.solve(Lf, b, zomplex=zomplex)
    assert np.linalg.norm(Af @ x.T - b.T) / np.linalg.norm(b) < TOL_RES
    assert np.linalg.norm(x - O.solve_complex(b)) / np.linalg.norm(x) < 1e-10
    # a real right-hand side against the complex factor: X is complex (cholmod_solve.c:1125-1134)
    xr = S.solve(Lf, b[0].real.copy())
    assert np.iscomplexobj(xr)
    assert np.linalg.norm(Af @ xr - b[0].real) / np.linalg.norm(b[0].real) < TOL_RES
    # forward and backward solves one by one: L y = P b, L^H z = y
    y = S.solve(Lf, b[0], sys=ch.SYS_L)
    Ld_ok = dense_check and n <= 2000
    if Ld_ok:
        assert np.linalg.norm(Ld @ y - b[0]) / np.linalg.norm(b[0]) < TOL_RES
        z = S.solve(Lf, b[0], sys=ch.SYS_Lt)
        assert np.linalg.norm(Ld.conj().T @ z - b[0]) / np.linalg.norm(b[0]) < TOL_RES
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()
    return err


def _check_not_posdef(use_gpu, quick):
    n, Ap, Ai, Ax = G.poisson3d(10)
    Ap, Ai, Ax = hermitian_from(n, Ap, Ai, Ax)
    perm = G.geometric_nd(10, 10, 10, 3)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    sup = O.super
    cand = [s for s in range(O.nsuper // 2, O.nsuper) if sup[s + 1] - sup[s] >= 6]
    kbad = int(sup[cand[0]] + 3)
    Ax2 = Ax.copy()
    Ax2[Ap[int(O.Perm[kbad])]] = -7.0                # the diagonal entry of that column
    assert O.factorize_complex(Ax2, quick_return=quick) == 1 and O.minor == kbad
    S = ch.Session(use_gpu=use_gpu)
    S.cm.quick_return_if_not_posdef = int(quick)
    A = S.sparse(n, Ap, Ai, Ax2, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1                   # TRUE, as the reference
    assert S.cm.status == ch.NOT_POSDEF
    fv = ch.FactorView(Lf)
    assert fv.minor == kbad
    mask = O.lower_mask()
    assert np.array_equal(fv.x[mask] != 0, O.xc[mask] != 0)
    assert np.linalg.norm((fv.x - O.xc)[mask]) / np.linalg.norm(O.xc[mask]) < TOL_L
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


def _check_real_factor_complex_rhs(use_gpu):
    """Real L, complex B: the real and imaginary parts are solved as 2 nrhs real
    right-hand sides (the reference's "dual" workspace, cholmod_solve.c:1553)."""
    n, Ap, Ai, Ax = G.poisson3d(8)
    perm = G.geometric_nd(8, 8, 8, 3)
    S = ch.Session(use_gpu=use_gpu)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    rng = np.random.default_rng(3)
    b = rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))
    for zomplex in (False, True):
        S.cm.prefer_zomplex = int(zomplex)
        x = S.solve(Lf, b, zomplex=zomplex)
        r = np.stack([G.sym_matvec(n, Ap, Ai, Ax, -1, x[q].real) + 1j * G.sym_matvec(n, Ap, Ai, Ax, -1, x[q].imag)
                      for q in range(3)]) - b
        assert np.linalg.norm(r) / np.linalg.norm(b) < TOL_RES
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


# ---- the oracle's complex template against numpy (no product code) --------------------

@pytest.mark.parametrize("name", ["bcsstk01", "p3d_9_nd", "box7r2_nd"])
@pytest.mark.parametrize("zomplex", [False, True])
def test_oracle_complex_is_cholesky_of_permuted_matrix(golden_dir, name, zomplex):
    n, Ap, Ai, Ax, perm = _case(name, golden_dir)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    assert O.factorize_complex(Ax, zomplex=zomplex) == 0
    Ld = dense_L(O, O.xc)
    P = O.Perm
    Ad = full_hermitian(n, Ap, Ai, Ax).toarray()[np.ix_(P, P)]
    ref = np.linalg.cholesky(Ad)
    assert np.linalg.norm(Ld - ref) / np.linalg.norm(ref) < 1e-13
    # same maps as the real analysis of the same pattern (same maps, different arithmetic)
    Or = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    for k in ("super", "pi", "px", "s"):
        assert np.array_equal(getattr(O, k), getattr(Or, k))


# ---- CPU path (Common->useGPU == 0) -------------------------------------------------------

@pytest.mark.parametrize("name", ["bcsstk01", "p3d_9_nd", "box7r2_nd", "p2d_40_nat"])
@pytest.mark.parametrize("zomplex", [False, True])
def test_complex_cpu_path_matches_oracle(golden_dir, name, zomplex):
    _check(name, golden_dir, use_gpu=0, zomplex=zomplex)


def test_complex_cpu_path_upper_stored_input(golden_dir):
    _check("p3d_9_nd", golden_dir, use_gpu=0, upper=True)
    _check("bcsstk01", golden_dir, use_gpu=0, upper=True, zomplex=True)


@pytest.mark.parametrize("quick", [False, True])
def test_complex_cpu_not_posdef_protocol(quick):
    _check_not_posdef(0, quick)


def test_real_factor_complex_rhs_cpu():
    _check_real_factor_complex_rhs(0)


def test_complex_type_mismatch_is_rejected(golden_dir):
    """A numeric real L cannot be refactorized with a complex A (and vice versa):
    reference cholmod_super_numeric.c:160-175."""
    n, Ap, Ai, Ax = G.poisson3d(5)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    cAp, cAi, cAx = hermitian_from(n, Ap, Ai, Ax)
    Ac = S.sparse(n, cAp, cAi, cAx, -1)
    Lf = S.analyze(A)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    assert S.factorize(Ac, Lf) == 0 and S.cm.status == ch.INVALID
    S.free_factor(Lf)
    S.free_sparse(A)
    S.free_sparse(Ac)
    assert S.cm.malloc_count == 0
    S.finish()


# ---- HIP engine ------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bcsstk01", "p3d_9_nd", "box7r2_nd", "p2d_40_nat", "p3d_24_nd", "box16r2_nd"])
@pytest.mark.parametrize("zomplex", [False, True])
def test_complex_gpu_matches_oracle(golden_dir, name, zomplex):
    assert ch.lib().cholmod_hip_probe() == 1, "no HIP device visible"
    _check(name, golden_dir, use_gpu=1, zomplex=zomplex)


@pytest.mark.gpu
def test_complex_gpu_upper_stored_and_device_resident(golden_dir):
    _check("p3d_9_nd", golden_dir, use_gpu=1, upper=True)
    # the factor stays in HBM (L->x NULL) until cholmod_l_factor_to_host; solves on the device
    n, Ap, Ai, Ax, perm = _case("p3d_24_nd", golden_dir)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    assert O.factorize_complex(Ax) == 0
    S = ch.Session(use_gpu=1, factor_on_device=True)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    assert not Lf.contents.x and Lf.contents.hip_on_device
    b = np.exp(1j * np.arange(n))
    x = S.solve(Lf, b)
    Af = full_hermitian(n, Ap, Ai, Ax)
    assert np.linalg.norm(Af @ x - b) / np.linalg.norm(b) < TOL_RES
    assert S.refactorize_resident(Lf, beta=0.5) == 1
    x2 = S.solve(Lf, b)
    assert np.linalg.norm(Af @ x2 + 0.5 * x2 - b) / np.linalg.norm(b) < TOL_RES
    assert S.refactorize_resident(Lf) == 1
    assert S.L.cholmod_l_factor_to_host(Lf, C.byref(S.cm)) == 1
    fv = ch.FactorView(Lf)
    mask = O.lower_mask()
    assert np.linalg.norm((fv.x - O.xc)[mask]) / np.linalg.norm(O.xc[mask]) < TOL_L
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


@pytest.mark.gpu
@pytest.mark.parametrize("quick", [False, True])
def test_complex_gpu_not_posdef_protocol(quick):
    _check_not_posdef(1, quick)


@pytest.mark.gpu
def test_real_factor_complex_rhs_gpu():
    _check_real_factor_complex_rhs(1)


# ---- the twin's update kernels: even-column contraction (CHOLMOD_HIP_PHI_TWIN) ------------------
# A complex multiply-add as four real ones (zherk / zgemm, t_cholmod_super_numeric.c:41-83,
# :682-717): the update kernels read the even columns of the embedded panels only and rebuild
# the 2 x 2 blocks in the lanes.  Every kernel that carries the variant, against the oracle's
# complex template, and the plain embedding (CHOLMOD_HIP_TWIN_FULL_K) beside it.

def test_twin_flag_is_checked_at_plan_creation():
    """A plan that claims the doubled structure of a twin and does not have it is refused
    (host-only plan: no device needed)."""
    lib = ch.lib()
    n, Ap, Ai, Ax = G.poisson3d(5)
    O = OracleFactor(n, Ap, Ai, -1, perm=None, postorder=True)
    sup, pi, px, s = (np.ascontiguousarray(getattr(O, k), dtype=np.int64) for k in ("super", "pi", "px", "s"))
    st = C.c_int(0)
    args = (n, len(sup) - 1, sup.ctypes.data_as(C.c_void_p), pi.ctypes.data_as(C.c_void_p),
            px.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p))
    P = lib.cholmod_hip_plan_create(*args, ch.HIP_PLAN_HOST_ONLY | ch.HIP_PHI_TWIN, C.byref(st))
    assert not P and st.value == ch.HIP_INVALID
    # the doubled structure passes
    sup2, pi2, px2 = 2 * sup, 2 * pi, 4 * px
    s2 = np.empty(2 * len(s), dtype=np.int64)
    s2[0::2] = 2 * s
    s2[1::2] = 2 * s + 1
    args2 = (2 * n, len(sup) - 1, sup2.ctypes.data_as(C.c_void_p), pi2.ctypes.data_as(C.c_void_p),
             px2.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p))
    P = lib.cholmod_hip_plan_create(*args2, ch.HIP_PLAN_HOST_ONLY | ch.HIP_PHI_TWIN, C.byref(st))
    assert P and st.value == 0
    lib.cholmod_hip_plan_destroy(P)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["default", "wave_tiles_everywhere", "four_wave_tiles_only", "no_fused_potrf", "no_fused_trsm", "wide_ob", "generic_kernels_only"])
@pytest.mark.parametrize("storage", ["complex_storage", "twin_even_columns", "plain_embedding"])
@pytest.mark.parametrize("name", ["p3d_24_nd", "box16r2_nd"])
def test_complex_gpu_twin_update_variants(golden_dir, monkeypatch, name, storage, variant):
    """The three forms the engine has for a complex factor -- its own storage (default on one GPU:
    CHOLMOD_HIP_CX_STORAGE, 2 xsize doubles, odd twin columns rebuilt in the kernels), the full twin with
    even-column updates, the plain embedding -- through every update / panel kernel variant."""
    assert ch.lib().cholmod_hip_probe() == 1, "no HIP device visible"
    if storage == "plain_embedding" and variant not in ("default", "wave_tiles_everywhere"):
        pytest.skip("the plain embedding runs the real kernels, covered by the real parity tests")
    kw = {}
    if storage == "twin_even_columns":
        monkeypatch.setenv("CHOLMOD_HIP_CX_TWIN", "1")
    elif storage == "plain_embedding":
        monkeypatch.setenv("CHOLMOD_HIP_TWIN_FULL_K", "1")
    if variant == "wave_tiles_everywhere":
        monkeypatch.setenv("CHOLMOD_HIP_UPD3_MIN_TILES", "1")       # k_update3 incl. partial tiles
    elif variant == "four_wave_tiles_only":
        monkeypatch.setenv("CHOLMOD_HIP_UPD3_MIN_TILES", "0")       # k_update2 / k_update2f only
    elif variant == "no_fused_potrf":
        kw["hip_flags"] = ch.HIP_NO_FUSED_POTRF
    elif variant == "no_fused_trsm":
        kw["hip_flags"] = ch.HIP_NO_FUSED_TRSM
    elif variant == "wide_ob":
        kw["hip_flags"] = ch.HIP_WIDE_OB
    elif variant == "generic_kernels_only":
        if storage != "complex_storage":
            pytest.skip("the thin-front kernel's complex form belongs to complex storage")
        monkeypatch.setenv("CHOLMOD_HIP_CX_NO_THIN", "1")           # (the default runs thin fronts through k_thin_front<..., CX>)
    _check(name, golden_dir, use_gpu=1, dense_check=False, session_kwargs=kw)


def _tiny_cases():
    """1 x 1, a dense 3 x 3 and a 70 x 70 arrow (one 70-row front wider than a tile) Hermitian positive definite matrices."""
    out = [(1, np.array([0, 1], dtype=np.int64), np.array([0], dtype=np.int64), np.array([2.5 + 0j]))]
    rng = np.random.default_rng(3)
    for n in (3, 70):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        A = M @ M.conj().T + n * np.eye(n)
        if n == 70:                     # arrow: dense last rows, diagonal elsewhere
            mask = np.zeros((n, n), dtype=bool)
            mask[np.diag_indices(n)] = True
            mask[-6:, :] = True
            A = np.where(mask | mask.T, A, 0)
            A[np.diag_indices(n)] = np.abs(A).sum(axis=1) + 1.0
        L = sp.tril(sp.csc_matrix(A)).tocsc()
        L.sort_indices()
        out.append((n, L.indptr.astype(np.int64), L.indices.astype(np.int64), L.data.astype(np.complex128)))
    return out


@pytest.mark.parametrize("use_gpu", [0, pytest.param(1, marks=pytest.mark.gpu)])
def test_complex_tiny_matrices(use_gpu, monkeypatch):
    for storage in (("cx",) if use_gpu == 0 else ("cx", "twin")):
        if storage == "twin":
            monkeypatch.setenv("CHOLMOD_HIP_CX_TWIN", "1")
        for (n, Ap, Ai, Ax) in _tiny_cases():
            S = ch.Session(use_gpu=use_gpu)
            A = S.sparse(n, Ap, Ai, Ax, -1)
            Lf = S.analyze(A)
            assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
            fv = ch.FactorView(Lf)
            Ld = dense_L(fv, fv.x)
            Af = full_hermitian(n, Ap, Ai, Ax).toarray()
            P = fv.Perm
            assert np.linalg.norm(Ld @ Ld.conj().T - Af[np.ix_(P, P)]) / np.linalg.norm(Af) < 1e-13
            assert np.all(np.diag(Ld).imag == 0)
            b = np.exp(1j * np.arange(n))
            x = S.solve(Lf, b)
            assert np.linalg.norm(Af @ x - b) / np.linalg.norm(b) < TOL_RES
            S.free_factor(Lf)
            S.free_sparse(A)
            assert S.cm.malloc_count == 0
            S.finish()


@pytest.mark.gpu
def test_complex_factor_moves_between_the_engine_and_the_cpu_path(golden_dir):
    """The same complex L factorized on the engine (complex storage), then with Common->useGPU = 0 (the CPU path
    needs the full twin: the engine-side twin is rebuilt), then on the engine again: the oracle's factor every time."""
    n, Ap, Ai, Ax, perm = _case("p3d_9_nd", golden_dir)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    assert O.factorize_complex(Ax) == 0
    mask = O.lower_mask()
    S = ch.Session(use_gpu=1)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    b = np.exp(1j * np.arange(n))
    Af = full_hermitian(n, Ap, Ai, Ax)
    for use_gpu, kind in ((1, 2), (0, 1), (1, 2)):
        S.cm.useGPU = use_gpu
        assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
        T = C.cast(Lf.contents.cx_twin, C.POINTER(ch.Factor))
        assert T.contents.hip_is_twin == kind
        fv = ch.FactorView(Lf)
        assert np.linalg.norm((fv.x - O.xc)[mask]) / np.linalg.norm(O.xc[mask]) < TOL_L
        x = S.solve(Lf, b)
        assert np.linalg.norm(Af @ x - b) / np.linalg.norm(b) < TOL_RES
    S.free_factor(Lf)
    S.free_sparse(A)
    assert S.cm.malloc_count == 0
    S.finish()


@pytest.mark.gpu
def test_complex_storage_is_half_the_twin(golden_dir, monkeypatch):
    """The engine's factor of a complex matrix occupies 2 xsize doubles in its own storage, 4 xsize as a twin
    (cholmod_hip_get_stats [5] = bytes of L on the device)."""
    n, Ap, Ai, Ax, perm = _case("p3d_24_nd", golden_dir)
    sizes, checks = {}, {}
    for twin in (False, True):
        if twin:
            monkeypatch.setenv("CHOLMOD_HIP_CX_TWIN", "1")
        S = ch.Session(use_gpu=1, factor_on_device=True)
        A = S.sparse(n, Ap, Ai, Ax, -1)
        Lf = S.analyze(A, perm)
        assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
        T = C.cast(Lf.contents.cx_twin, C.POINTER(ch.Factor))
        st = np.zeros(ch.CHOLMOD_HIP_NSTATS)
        S.L.cholmod_hip_get_stats(T.contents.hip_plan, st.ctypes.data)
        sizes[twin] = (st[5], int(Lf.contents.xsize), int(T.contents.hip_is_twin))
        # the factor invariants are those of the twin the storage stands for (k_factor_checks rebuilds the odd columns)
        out = np.zeros(5)
        assert S.L.cholmod_hip_factor_checks(T.contents.hip_plan, out.ctypes.data) == 0
        checks[twin] = out.copy()
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()
    assert sizes[False][2] == 2 and sizes[True][2] == 1
    a, b = checks[False], checks[True]
    assert abs(a[0] - b[0]) < 1e-11 * abs(b[0]) and abs(a[3] - b[3]) < 1e-11 * b[3] and a[1] == b[1] == 0 and a[2] == b[2] == 0 and a[4] == b[4] == 0, (a, b)
    assert sizes[False][0] == 8.0 * 2 * sizes[False][1]
    assert sizes[True][0] == 8.0 * 4 * sizes[True][1]


@pytest.mark.gpu
def test_complex_storage_thin_fronts_take_the_thin_kernel(golden_dir, monkeypatch):
    """Complex storage runs the fronts of up to 136 twin rows through the LDS-resident thin-front kernel in its complex
    form (cholmod_hip_get_stats [21] = fronts handled there); CHOLMOD_HIP_CX_NO_THIN=1 sends every front to the generic
    kernels, as before round 4.  Both against the oracle (a 2D problem: nearly every front is thin)."""
    n, Ap, Ai, Ax, perm = _case("p2d_40_nat", golden_dir)
    seen = {}
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("CHOLMOD_HIP_CX_NO_THIN", "1")
        S = ch.Session(use_gpu=1, factor_on_device=True)
        A = S.sparse(n, Ap, Ai, Ax, -1)
        Lf = S.analyze(A, perm)
        assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
        T = C.cast(Lf.contents.cx_twin, C.POINTER(ch.Factor))
        st = np.zeros(ch.CHOLMOD_HIP_NSTATS)
        S.L.cholmod_hip_get_stats(T.contents.hip_plan, st.ctypes.data)
        seen[generic] = (int(st[21]), int(T.contents.hip_is_twin))
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()
    assert seen[False][1] == 2 and seen[True][1] == 2
    assert seen[False][0] > 0 and seen[True][0] == 0, seen
    monkeypatch.delenv("CHOLMOD_HIP_CX_NO_THIN")
    _check("p2d_40_nat", golden_dir, use_gpu=1, dense_check=False)


@pytest.mark.gpu
def test_complex_gpu_big_fronts_even_column_updates(monkeypatch):
    """A complex problem whose top fronts reach the one-wave-per-tile update kernel by themselves
    (twin root of 2 x 1600 columns), by residual, factor invariants and against the plain embedding,
    in the engine's three forms."""
    m = 40
    n, Ap, Ai, Ax = G.poisson3d(m)
    perm = G.geometric_nd(m, m, m, 4)
    cAp, cAi, cAx = hermitian_from(n, Ap, Ai, Ax)
    Af = full_hermitian(n, cAp, cAi, cAx)
    b = np.exp(1j * np.arange(n))
    xs = {}
    for form, env in (("complex_storage", None), ("twin_even_columns", "CHOLMOD_HIP_CX_TWIN"), ("plain_embedding", "CHOLMOD_HIP_TWIN_FULL_K")):
        monkeypatch.delenv("CHOLMOD_HIP_CX_TWIN", raising=False)
        monkeypatch.delenv("CHOLMOD_HIP_TWIN_FULL_K", raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        S = ch.Session(use_gpu=1)
        A = S.sparse(n, cAp, cAi, cAx, -1)
        Lf = S.analyze(A, perm)
        assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
        assert S.L.cholmod_l_check_factor(Lf, C.byref(S.cm)) == 1
        fv = ch.FactorView(Lf)
        xs[form] = fv.x.copy()
        x = S.solve(Lf, b)
        assert np.linalg.norm(Af @ x - b) / np.linalg.norm(b) < TOL_RES
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()
    ref = xs["plain_embedding"]
    for form in ("complex_storage", "twin_even_columns"):
        assert np.linalg.norm(xs[form] - ref) / np.linalg.norm(ref) < TOL_L, form
