"""Randomised parity run against the oracle (GPU box): random sparse SPD patterns,
random / built-in / natural orderings, random not-positive-definite injections,
unpacked inputs, 1-3 right-hand sides; every fourth case with complex or zomplex Hermitian
values (random phases on the off-diagonal entries) against the oracle's complex template.
Prints one line per failure and a summary."""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import OracleFactor
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G


def rand_spd(rng, n, density, band):
    R = sp.random(n, n, density=min(1.0, density), random_state=int(rng.integers(1 << 30)), format="coo")
    if band:
        keep = np.abs(R.row - R.col) < max(2, n // 8)
        R = sp.coo_matrix((R.data[keep], (R.row[keep], R.col[keep])), shape=(n, n))
    R = R.tocsr()
    A = (R + R.T).tocsr()
    A.data[:] = -np.abs(A.data) - 0.05
    A = A + sp.diags(np.asarray(-A.sum(axis=1)).ravel() + rng.uniform(0.1, 2.0))
    T = sp.tril(A).tocsc()
    T.sort_indices()
    return T.indptr.astype(np.int64), T.indices.astype(np.int64), T.data.astype(np.float64)


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(ncase):
        kind = it % 6
        if kind == 0:
            m = int(rng.integers(5, 22)); n, Ap, Ai, Ax = G.poisson3d(m, int(rng.integers(3, 20)), int(rng.integers(2, 15)))
        elif kind == 1:
            m = int(rng.integers(10, 120)); n, Ap, Ai, Ax = G.poisson2d(m, int(rng.integers(5, 90)))
        elif kind == 2:
            n = int(rng.integers(700, 2200)); Ap, Ai, Ax = rand_spd(rng, n, 30.0 / n, True)       # banded, fat supernodes
        else:
            n = int(rng.integers(2, 3000)); Ap, Ai, Ax = rand_spd(rng, n, rng.uniform(1.0, 8.0) / n, bool(rng.integers(2)))
        omode = ("natural", "nesdis", "random")[int(rng.integers(3))]
        perm = rng.permutation(n).astype(np.int64) if omode == "random" else None
        flags = int(rng.choice([0, 0, 0, 64, 128, 16, 2048, 512, 8192, 8192 | 128]))      # (8192: the 256-column panel chain)
        cx = (it % 4 == 3)
        zomplex = cx and bool(rng.integers(2))
        if cx:
            Ax = Ax.astype(np.complex128)
            cols = np.repeat(np.arange(n), np.diff(Ap))
            off = Ai != cols
            Ax[off] *= np.exp(1j * rng.uniform(0, 2 * np.pi, int(off.sum())))
        S = ch.Session(ordering=omode if omode != "random" else "natural", hip_flags=flags,
                       use_gpu=int(os.environ.get("FUZZ_USE_GPU", "1")))      # 0: the product's CPU path
        A = S.sparse(n, Ap, Ai, Ax, -1, zomplex=zomplex)
        Lf = S.analyze(A, perm)
        fv = ch.FactorView(Lf)
        Pfinal = fv.Perm.copy()
        inject = rng.random() < 0.25
        Axx = Ax.copy()
        if inject:
            k = int(rng.integers(n))
            Axx[Ap[k]] = -abs(Axx[Ap[k]])            # negative diagonal somewhere
            S.free_sparse(A)
            A = S.sparse(n, Ap, Ai, Axx, -1, zomplex=zomplex)
        ok = S.factorize(A, Lf)
        O = OracleFactor(n, Ap, Ai, -1, perm=Pfinal, postorder=True)
        st = O.factorize_complex(Axx, zomplex=zomplex) if cx else O.factorize(Axx)
        Ox = O.xc if cx else O.x
        fv = ch.FactorView(Lf)
        msg = []
        for key in ("Perm", "super", "pi", "px", "s"):
            if not np.array_equal(getattr(fv, key), getattr(O, key)):
                msg.append("map " + key)
        if ok != 1 or (S.cm.status != 0) != (st != 0):
            msg.append(f"status gpu {S.cm.status} oracle {st}")
        if st != 0 and fv.minor != O.minor:
            msg.append(f"minor {fv.minor} vs {O.minor}")
        mk = O.lower_mask()
        den = np.linalg.norm(Ox[mk])
        err = np.linalg.norm((fv.x - Ox)[mk]) / (den if den > 0 else 1.0)
        if not (err < 1e-11):
            msg.append(f"L err {err:.2e}")
        if st == 0:
            nr = int(rng.integers(1, 4))
            b = rng.standard_normal((nr, n)) if nr > 1 else rng.standard_normal(n)
            if cx:
                b = b + 1j * rng.standard_normal(b.shape)
                Al = sp.csc_matrix((Axx, Ai, Ap), shape=(n, n))
                Af = (Al + sp.tril(Al, -1).conj().T).tocsr()
            x = S.solve(Lf, b, zomplex=zomplex)
            for bb, xx in zip(np.atleast_2d(b), np.atleast_2d(x)):
                r = (Af @ xx if cx else G.sym_matvec(n, Ap, Ai, Axx, -1, xx)) - bb
                if not (np.linalg.norm(r) <= 1e-9 * np.linalg.norm(bb)):
                    msg.append(f"resid {np.linalg.norm(r) / np.linalg.norm(bb):.2e}")
        if st == 0 and not cx:
            # the same pattern with new values through cholmod_l_factorize again: the values-only
            # path -- A through the assembly map of the resident S (thin fronts `mapped`, leaf
            # fronts two to a wave), sometimes with a failing pivot
            import ctypes as C
            Ax2 = Axx * (1.0 + 0.25 * rng.random())
            diag = Ai == np.repeat(np.arange(n), np.diff(Ap))
            Ax2[diag] += rng.uniform(0.0, 1.0, int(diag.sum()))
            if rng.random() < 0.2:
                k = int(rng.integers(n))
                Ax2[Ap[k]] = -abs(Ax2[Ap[k]])
            ch._view(A.contents.x, len(Ax2), C.c_double, np.float64)[:] = Ax2
            ok2 = S.factorize(A, Lf)
            st2 = O.factorize(Ax2)
            fv = ch.FactorView(Lf)
            if ok2 != 1 or (S.cm.status != 0) != (st2 != 0):
                msg.append(f"again: status gpu {S.cm.status} oracle {st2}")
            if st2 != 0 and fv.minor != O.minor:
                msg.append(f"again: minor {fv.minor} vs {O.minor}")
            den = np.linalg.norm(O.x[mk])
            err2 = np.linalg.norm((fv.x - O.x)[mk]) / (den if den > 0 else 1.0)
            if not (err2 < 1e-11):
                msg.append(f"again: L err {err2:.2e}")
        if msg:
            bad += 1
            print(f"FAIL case {it} kind {kind} n {n} order {omode} flags {flags} inject {inject} complex {cx} zomplex {zomplex}: "
                  + "; ".join(msg), flush=True)
        S.free_factor(Lf)
        S.free_sparse(A)
        if S.cm.malloc_count != 0:
            print(f"LEAK case {it}: malloc_count {S.cm.malloc_count}")
            bad += 1
        S.finish()
    print(f"fuzz: {ncase} cases, {bad} failures (seed {seed})")


if __name__ == "__main__":
    main()
