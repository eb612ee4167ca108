"""Reference-HELD expected values for the symbolic part of the path: LDL/Demo/ldlmain.out records, for each of the 24 valid
matrices of LDL/Matrix (copied as data to tests/golden/ldl/), the number of entries of L and the flop count of a factorization
under the permutation stored in the file and under the natural order (LDL/Demo/ldlmain.c:285-297:
flops = sum Lnz (Lnz + 2), Lnz = column count without the diagonal).  The elimination tree and the column counts behind them are
the ones CHOLMOD's analysis computes (cholmod_rowcolcounts.c: Common->lnz = sum c, Common->fl = sum c^2 with c = Lnz + 1), so

    Common->lnz - n == "Nz in L",       Common->fl - n == "Flop count"

must hold for the oracle and for the product under the same permutation -- `Common->fl` is the numerator of the metric
(SURVEY 8d).  The six invalid files (A25 .. A30) must be refused; the matrices a Cholesky factorization exists for are pushed
through the supernodal path under the file's permutation (CPU path here, HIP path under -m gpu) and compared with the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from ldl_files import read_ldl_matrix, recorded, upper_csc
from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch

HERE = os.path.dirname(os.path.abspath(__file__))
LDL = os.path.join(HERE, "golden", "ldl")
REC = recorded(os.path.join(LDL, "ldlmain.out"))
VALID = sorted(k for k, v in REC.items() if not v.get("invalid"))
INVALID = sorted(k for k, v in REC.items() if v.get("invalid"))


def test_the_recorded_file_is_complete():
    assert len(REC) == 30 and len(VALID) == 24 and INVALID == ["A25", "A26", "A27", "A28", "A29", "A30"]
    assert REC["A13"]["given"] == (441, 5961.0) and REC["A13"]["natural"] == (829, 20103.0)      # HB/bcsstk01
    assert REC["A21"]["given"] == (2145, 97955.0)                                                # HB/bcsstk02: dense


@pytest.mark.parametrize("name", VALID)
def test_oracle_counts_match_the_recorded_output(name):
    m = read_ldl_matrix(os.path.join(LDL, name))
    n = m["n"]
    if n == 0:
        assert REC[name]["given"] == (0, 0.0) and REC[name]["natural"] == (0, 0.0)
        return
    Ap, Ai, _ = upper_csc(m)
    for key, perm in (("given", m["P"]), ("natural", np.arange(n))):
        for postorder in (False, True):          # a postordering moves columns, never the counts' sums
            L = OracleFactor(n, Ap, Ai, 1, perm=perm, postorder=postorder)
            assert (int(L.lnz) - n, float(L.fl) - n) == REC[name][key], (name, key, postorder)


@pytest.mark.parametrize("name", VALID)
def test_product_counts_match_the_recorded_output(name):
    m = read_ldl_matrix(os.path.join(LDL, name))
    n = m["n"]
    if n == 0:
        return
    Ap, Ai, Ax = upper_csc(m)
    for relax in (False, True):
        S = ch.Session(use_gpu=0, ordering="natural")
        if not relax:
            for i in range(3):
                S.cm.nrelax[i] = 0
                S.cm.zrelax[i] = 0.0
        A = S.sparse(n, Ap, Ai, Ax, 1)
        for key, perm in (("given", m["P"]), ("natural", None)):
            Lf = S.analyze(A, perm=perm)
            assert (int(S.cm.lnz) - n, float(S.cm.fl) - n) == REC[name][key], (name, key)
            cc = ch.FactorView(Lf).ColCount
            assert int(cc.sum()) - n == REC[name][key][0]
            assert float((cc.astype(np.float64) ** 2).sum()) - n == REC[name][key][1]
            S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()


@pytest.mark.parametrize("name", INVALID)
def test_invalid_files_are_refused(name):
    """ldlmain.out: "invalid matrix and/or permutation" (LDL_valid_matrix / LDL_valid_perm) -- here cholmod_l_check_sparse and
    the permutation check of cholmod_l_analyze_p (cholmod_analyze.c:617-635)."""
    m = read_ldl_matrix(os.path.join(LDL, name))
    n = m["n"]
    S = ch.Session(use_gpu=0, ordering="natural")
    Ap, Ai, Ax = m["Ap"], m["Ai"], m["Ax"]
    bad_matrix = "invalid perm" not in REC[name]["name"]
    if bad_matrix:
        nz = len(Ai)
        A = S.L.cholmod_l_allocate_sparse(n, n, max(nz, 1), 1, 1, 0, ch.REAL, C.byref(S.cm))
        a = A.contents
        ch._view(a.p, n + 1, C.c_int64, np.int64)[:] = Ap[:n + 1]
        if nz:
            ch._view(a.i, nz, C.c_int64, np.int64)[:] = Ai
            ch._view(a.x, nz, C.c_double, np.float64)[:] = Ax if len(Ax) == nz else 1.0
        assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 0
        assert S.cm.status == ch.INVALID
        S.free_sparse(A)
    else:
        import scipy.sparse as sp
        U = sp.triu(sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)), format="csc")
        A = S.sparse(n, U.indptr.astype(np.int64), U.indices.astype(np.int64), U.data, 1)
        assert S.L.cholmod_l_check_sparse(A, C.byref(S.cm)) == 1
        with pytest.raises(RuntimeError):
            S.analyze(A, perm=m["P"])
        assert S.cm.status == ch.INVALID
        S.free_sparse(A)
    S.finish()


def _factor_case(name, use_gpu):
    m = read_ldl_matrix(os.path.join(LDL, name))
    n = m["n"]
    Ap, Ai, Ax = upper_csc(m)
    O = OracleFactor(n, Ap, Ai, 1, perm=m["P"], postorder=True)
    info = O.factorize(Ax)
    S = ch.Session(use_gpu=use_gpu, ordering="natural")
    A = S.sparse(n, Ap, Ai, Ax, 1)
    Lf = S.analyze(A, perm=m["P"])
    assert S.factorize(A, Lf) == 1
    fv = ch.FactorView(Lf)
    assert bool(fv.is_super)
    for key in ("super", "pi", "px", "s", "Perm"):
        assert np.array_equal(getattr(fv, key), getattr(O, key)), key
    assert int(Lf.contents.minor) == int(O.minor)
    if info == 0:
        assert S.cm.status == ch.OK
        mask = O.lower_mask()       # (the lower trapezoids: what LAPACK leaves above a diagonal block is not part of L)
        assert np.linalg.norm((fv.x - O.x)[mask]) <= 1e-12 * np.linalg.norm(O.x[mask])
        assert not np.any(fv.x[~mask])
        rng = np.random.default_rng(7)
        b = rng.standard_normal(n)
        x = S.solve(Lf, b)
        import scipy.sparse as sp
        U = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
        Af = U + sp.triu(U, 1).T
        assert np.abs(Af @ x - b).max() <= 1e-9 * max(1.0, np.abs(Af).sum(axis=1).max() * np.abs(x).max())
    else:
        assert S.cm.status == ch.NOT_POSDEF
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()
    return info


NONEMPTY = [k for k in VALID if REC[k]["name"].split("n: ")[1].split()[0] != "0"]


@pytest.mark.parametrize("name", NONEMPTY)
def test_cpu_path_on_the_ldl_matrices(name):
    _factor_case(name, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NONEMPTY)
def test_hip_path_on_the_ldl_matrices(name):
    _factor_case(name, 1)


def test_some_of_them_are_positive_definite():
    """(so that the numeric comparison above is not vacuous: bcsstk01 / bcsstk02 / mesh1e1 and their jumbled twins at least)"""
    spd = [k for k in NONEMPTY if _factor_case(k, 0) == 0]
    for k in ("A13", "A14", "A17", "A18", "A21", "A22"):
        assert k in spd
