"""The product's CPU supernodal path (Common->useGPU == 0; reference
CHOLMOD/Supernodal/t_cholmod_super_numeric.c:183-192, BASELINE.json configs[0]):
suitesparse_amd/csrc/host/cpu_numeric.c against the oracle -- maps bit-exact,
factor to 1e-12, residual, the not-positive-definite protocol, the demo driver
on bcsstk01.  Runs without a GPU, with the dlopen'ed BLAS if one is found and
with the built-in C kernels."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_L = 1e-12


def _case(name, golden_dir):
    if name == "bcsstk01":
        rec = json.load(open(os.path.join(golden_dir, "reference_recorded.json")))["bcsstk01"]
        return G.read_triplet(os.path.join(golden_dir, "bcsstk01.tri")) + (np.array(rec["Perm"]),)
    if name == "bcsstk02":
        return G.read_triplet(os.path.join(golden_dir, "bcsstk02.tri")) + (None,)
    if name == "p3d_14_nd":
        return G.poisson3d(14) + (-1, G.geometric_nd(14, 14, 14, 4))
    if name == "p2d_70_nd":
        return G.poisson2d(70) + (-1, G.geometric_nd(70, 70, 1, 4))
    if name == "box8r2_nd":
        return G.box_stencil3d(8, 2) + (-1, G.geometric_nd(8, 8, 8, 3))
    if name == "p3d_9x7x11_nat":
        return G.poisson3d(9, 7, 11) + (-1, None)
    raise KeyError(name)


def _scipy_blas():
    """LP64 OpenBLAS shipped with scipy, if any (CHOLMOD_BLAS_LIBRARY syntax path:prefix)."""
    import glob
    try:
        import scipy
    except Exception:
        return None
    d = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
    c = glob.glob(os.path.join(d, "libscipy_openblas-*.so"))
    return c[0] + ":scipy_" if c else None


def _run_child(code, env_extra):
    """The BLAS binding is per process: run the check in a child with the wanted
    CHOLMOD_BLAS_LIBRARY."""
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


CHECK = r'''
import sys, json, os
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch, generators as G
from test_cpu_path import _case
name = sys.argv[1] if len(sys.argv) > 1 else os.environ["CASE"]
n, Ap, Ai, Ax, stype, perm = _case(name, "tests/golden")
S = ch.Session(use_gpu=0)
A = S.sparse(n, Ap, Ai, Ax, stype)
Lf = S.analyze(A, perm)
assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK, S.cm.status
fv = ch.FactorView(Lf)
assert fv.x is not None and not Lf.contents.hip_plan        # host factor, no engine involved
O = OracleFactor(n, Ap, Ai, stype, perm=perm, postorder=True)
assert O.factorize(Ax) == 0
for k in ("Perm", "ColCount", "super", "pi", "px", "s"):
    assert np.array_equal(getattr(fv, k), getattr(O, k)), k
m = O.lower_mask()
err = np.linalg.norm((fv.x - O.x)[m]) / np.linalg.norm(O.x[m])
assert err < 1e-12, err
assert np.all(fv.x[~m] == 0)
assert S.L.cholmod_l_check_factor(Lf, __import__("ctypes").byref(S.cm)) == 1
b = G.demo_rhs(n)
x = S.solve(Lf, b)
r = G.sym_matvec(n, Ap, Ai, Ax, stype, x) - b
assert np.linalg.norm(r) / np.linalg.norm(b) < 1e-11
rng = np.random.default_rng(1)
B = rng.standard_normal((3, n))
assert np.linalg.norm(S.solve(Lf, B, ch.SYS_L) - O.lsolve(B)) < 1e-11 * np.linalg.norm(B)
assert np.linalg.norm(S.solve(Lf, B, ch.SYS_Lt) - O.ltsolve(B)) < 1e-9 * np.linalg.norm(O.ltsolve(B))
assert S.cm.cholmod_cpu_potrf_calls == fv.nsuper and S.cm.cholmod_gpu_potrf_calls == 0
S.free_factor(Lf); S.free_sparse(A)
assert S.cm.malloc_count == 0
S.finish()
print("ok", name, err)
'''


@pytest.mark.parametrize("blas", ["builtin", "dlopen"])
@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk02", "p3d_14_nd", "p2d_70_nd", "box8r2_nd", "p3d_9x7x11_nat"])
def test_cpu_factor_and_solves_match_oracle(name, blas):
    lib = "none" if blas == "builtin" else _scipy_blas()
    if lib is None:
        pytest.skip("no LP64 BLAS with LAPACK found for dlopen")
    out = _run_child(CHECK, {"CHOLMOD_BLAS_LIBRARY": lib, "CASE": name})
    assert out.startswith("ok")


@pytest.mark.parametrize("quick", [False, True])
def test_cpu_not_posdef_protocol_matches_oracle(quick):
    n, Ap, Ai, Ax = G.poisson3d(10)
    perm = G.geometric_nd(10, 10, 10, 3)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    sup = O.super
    cand = [s for s in range(O.nsuper // 2, O.nsuper) if sup[s + 1] - sup[s] >= 6]
    kbad = int(sup[cand[0]] + 3)
    Ax2 = Ax.copy()
    Ax2[Ap[int(O.Perm[kbad])]] = -7.0
    assert O.factorize(Ax2, quick_return=quick) == 1 and O.minor == kbad
    S = ch.Session(use_gpu=0)
    S.cm.quick_return_if_not_posdef = int(quick)
    A = S.sparse(n, Ap, Ai, Ax2, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1               # TRUE, as the reference
    assert S.cm.status == ch.NOT_POSDEF
    fv = ch.FactorView(Lf)
    assert fv.minor == kbad
    mask = O.lower_mask()
    assert np.array_equal(fv.x[mask] != 0, (O.x != 0)[mask])
    assert np.linalg.norm((fv.x - O.x)[mask]) / np.linalg.norm(O.x[mask]) < TOL_L
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


PAR_CHECK = r'''
import sys, os, ctypes
import numpy as np
sys.path.insert(0, ".")
from oracle.oracle import OracleFactor, bind_blas
from suitesparse_amd import cholmod as ch, generators as G
m = int(os.environ["GRID"])
n, Ap, Ai, Ax = G.poisson3d(m)
perm = G.geometric_nd(m, m, m, 4)
bind_blas()
O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
mask = O.lower_mask()
lib = os.environ.get("CHOLMOD_BLAS_LIBRARY", "none")
getter = None
if lib != "none":
    path, prefix = lib.split(":")
    B = ctypes.CDLL(path)
    getter = getattr(B, prefix + "openblas_get_num_threads")
    # the caller's setting must survive the factorization (not above the library's own maximum, OMP_NUM_THREADS at its load)
    want = min(2, int(os.environ.get("OMP_NUM_THREADS", "1")))
    getattr(B, prefix + "openblas_set_num_threads")(want)
S = ch.Session(use_gpu=0)
# a failing pivot deep in a subtree (phase A), in a supernode of the top part (phase B: the root), and none at all
sup = O.super
nsc = np.diff(sup)
cases = [None, int(sup[O.nsuper // 3] + min(2, nsc[O.nsuper // 3] - 1)), int(sup[O.nsuper - 1] + nsc[O.nsuper - 1] // 2)]
for kbad in cases:
    vals = Ax.copy()
    if kbad is not None:
        vals[Ap[int(O.Perm[kbad])]] = -7.0
    st = O.factorize(vals)
    A = S.sparse(n, Ap, Ai, vals, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1
    fv = ch.FactorView(Lf)
    assert S.cm.status == (ch.NOT_POSDEF if st == 1 else ch.OK) and fv.minor == O.minor, (kbad, S.cm.status, fv.minor, O.minor)
    assert np.array_equal(fv.x[mask] != 0, (O.x != 0)[mask]), kbad
    err = np.linalg.norm((fv.x - O.x)[mask]) / np.linalg.norm(O.x[mask])
    assert err < 1e-12, (kbad, err)
    assert np.all(fv.x[~mask] == 0)
    if kbad is None:
        b = G.demo_rhs(n)
        x = S.solve(Lf, b)
        assert np.linalg.norm(G.sym_matvec(n, Ap, Ai, vals, -1, x) - b) / np.linalg.norm(b) < 1e-11
        assert S.cm.cholmod_cpu_potrf_calls >= fv.nsuper
    S.free_factor(Lf); S.free_sparse(A)
    assert S.cm.malloc_count == 0
if getter is not None:
    assert getter() == want, getter()
S.finish()
print("ok")
'''


@pytest.mark.parametrize("blas", ["builtin", "dlopen"])
@pytest.mark.parametrize("threads,grid", [(1, 24), (8, 24), (8, 40), (3, 32)])
def test_cpu_path_on_several_threads(blas, threads, grid):
    """Round 6: independent subtrees on one thread each, the top supernodes by tiles (cpu_numeric.c), the BLAS on one
    thread per call and its entry thread count restored.  Poisson 40^3 has a 1 600-column root and supernodes beyond the
    tiling threshold; against the oracle, with a failing pivot in a subtree, one in the root, and none."""
    lib = "none" if blas == "builtin" else _scipy_blas()
    if lib is None:
        pytest.skip("no LP64 BLAS with LAPACK found for dlopen")
    if blas == "builtin" and grid > 32:
        pytest.skip("the built-in kernels are for plumbing, not for 40^3")
    out = _run_child(PAR_CHECK, {"CHOLMOD_BLAS_LIBRARY": lib, "GRID": str(grid), "OMP_NUM_THREADS": str(threads)})
    assert out.strip().endswith("ok")


def test_a_blas_with_64_bit_integers_is_refused_without_being_called(tmp_path):
    """An ILP64 library found under the LP64 names is recognised by inspection -- ilaver_ writes eight bytes per integer --
    and not used: none of its compute entry points may be called (here they abort the process, as a reference xerbla
    that STOPs would), the factorization falls back to the built-in kernels (round-5 advisor)."""
    src = tmp_path / "fake_ilp64.c"
    src.write_text(r'''
#include <stdlib.h>
void ilaver_ (long *a, long *b, long *c) { *a = 3 ; *b = 12 ; *c = 0 ; }
void dgemm_ (void) { abort () ; }
void dsyrk_ (void) { abort () ; }
void dtrsm_ (void) { abort () ; }
void dpotrf_ (void) { abort () ; }
''')
    so = tmp_path / "libfake_ilp64.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    code = r'''
import sys, ctypes
sys.path.insert(0, ".")
from suitesparse_amd import cholmod as ch, generators as G
n, Ap, Ai, Ax = G.poisson2d(12)
S = ch.Session(use_gpu=0)
A = S.sparse(n, Ap, Ai, Ax, -1)
Lf = S.analyze(A)
assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
S.L.ssamd_cpu_blas_name.restype = ctypes.c_char_p
print("blas:", S.L.ssamd_cpu_blas_name().decode())
'''
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, CHOLMOD_BLAS_LIBRARY=str(so)),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fails the LP64 self-check" in out.stderr
    assert "libfake_ilp64" not in out.stdout.split("blas:")[1]


def test_env_default_is_cpu_like_the_reference(monkeypatch):
    """Common->useGPU == -1 (cholmod_l_start): CHOLMOD_USE_GPU unset selects the CPU
    (CHOLMOD/Supernodal/cholmod_super_symbolic.c:286-291), =1 the GPU."""
    n, Ap, Ai, Ax = G.poisson2d(8)
    for env, want in ((None, 0), ("0", 0), ("1", 1)):
        if env is None:
            monkeypatch.delenv("CHOLMOD_USE_GPU", raising=False)
        else:
            monkeypatch.setenv("CHOLMOD_USE_GPU", env)
        S = ch.Session(use_gpu=-1)
        A = S.sparse(n, Ap, Ai, Ax, -1)
        Lf = S.analyze(A)
        assert S.cm.useGPU == want
        if want == 0:
            assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK and not Lf.contents.hip_plan
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()


def _check_demo_output(out):
    """What examples/cholmod_l_demo.c prints on bcsstk01 with the reference's recorded permutation: the reference's symbolic
    profile, three solve methods and the refinement step with residuals at rounding level, rcond as numpy has
    it for the same factor, nothing left allocated."""
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    assert "7 supernodes, ssize 101, xsize 1064, maxcsize 169, maxesize 13" in txt, txt
    assert "Analyze: flop 6009 lnz 489" in txt, txt
    assert "minor 48, status 0" in txt, txt
    assert "ints in L:             221, doubles in L:            1064" in txt or "doubles in L:            1064" in txt, txt
    res = [float(v) for v in txt.split("residual (|Ax-b|/(|A||x|+|b|)):")[1].split("\n")[0].split()]
    assert len(res) == 3 and all(0 <= r < 1e-12 for r in res), res
    assert float(txt.split("residual ")[-1].split()[0]) < 1e-12 and "after iterative refinement" in txt
    assert "solve2  walltime" in txt and "(100 trials)" in txt
    rc = float(txt.split("rcond")[1].split()[0])
    assert 1e-7 < rc < 1e-4, rc                       # ((min L_jj / max L_jj)^2 of bcsstk01: 5.7e-06 under this ordering family)
    assert "malloc_count 0 memory_inuse 0" in txt


def test_c_demo_on_cpu_path(golden_dir, tmp_path):
    """BASELINE.json configs[0]: the demo flow on bcsstk01 with Common->useGPU = 0."""
    rec = json.load(open(os.path.join(golden_dir, "reference_recorded.json")))["bcsstk01"]
    exe = str(tmp_path / "cholmod_l_demo")
    lib = os.path.join(ROOT, "suitesparse_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cholmod_l_demo.c"), "-L", lib, "-lcholmod_amd",
                           f"-Wl,-rpath,{lib}", "-lm", "-o", exe])
    permfile = tmp_path / "perm.txt"
    permfile.write_text(" ".join(str(v) for v in rec["Perm"]))
    with open(os.path.join(golden_dir, "bcsstk01.tri")) as f:
        out = subprocess.run([exe, "-cpu", "-perm", str(permfile)], stdin=f, capture_output=True, text=True, timeout=300)
    _check_demo_output(out)


def test_c_demo_on_an_unsymmetric_matrix(tmp_path):
    """The demo driver on a rectangular A (stype 0): it factorizes A*A' + 1e-6 I as the reference driver does
    (CHOLMOD/Demo/cholmod_l_demo.c:280-286) and checks (A*A' + beta*I) x = b (:537-575)."""
    import scipy.sparse as sp
    m, n = 40, 70
    M = sp.random(m, n, density=0.08, random_state=1, format="coo")
    M = (M + sp.coo_matrix((np.ones(m), (np.arange(m), np.arange(m))), shape=(m, n))).tocoo()
    f = tmp_path / "rect.tri"
    f.write_text(f"{m} {n} {M.nnz} 0\n" + "".join(f"{i} {j} {v!r}\n" for i, j, v in zip(M.row, M.col, M.data)))
    exe = str(tmp_path / "cholmod_l_demo")
    lib = os.path.join(ROOT, "suitesparse_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cholmod_l_demo.c"), "-L", lib, "-lcholmod_amd",
                           f"-Wl,-rpath,{lib}", "-lm", "-o", exe])
    out = subprocess.run([exe, "-cpu", str(f)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    assert "A: 40-by-70" in txt and "Factorizing A*A'+beta*I" in txt and "minor 40, status 0" in txt, txt
    res = [float(v) for v in txt.split("residual (|Ax-b|/(|A||x|+|b|)):")[1].split("\n")[0].split()]
    assert len(res) == 3 and all(0 <= r < 1e-12 for r in res), res
    assert "malloc_count 0 memory_inuse 0" in txt
