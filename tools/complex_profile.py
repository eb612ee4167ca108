"""Per-kind device times of a complex (Hermitian) factorization in complex storage, next to the real one of the same pattern.
usage: complex_profile.py [m=64]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from suitesparse_amd import cholmod as ch, generators as G
KIND = {0: "zero", 1: "extend_add", 2: "potrf", 3: "trsm", 5: "update64", 8: "thin", 9: "update+potrf", 10: "trsm+upd+potrf", 12: "update_w", 16: "chain256f"}
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, Ap, Ai, Ax = G.poisson3d(m)
perm = G.geometric_nd(m, m, m, 4)
Az = G.hermitian_phases(n, Ap, Ai, Ax, seed=m)
for tag, vals in (("real", Ax), ("complex", Az)):
    S = ch.Session(factor_on_device=True, ordering="default")
    A = S.sparse(n, Ap, Ai, vals, -1)
    Lf = S.analyze(A, perm)
    assert S.factorize(A, Lf) == 1
    assert S.refactorize_resident(Lf) == 1
    T = C.cast(Lf.contents.cx_twin, C.POINTER(ch.Factor)) if tag == "complex" else Lf
    S.set_profiling(T, True)
    assert S.refactorize_resident(Lf) == 1
    p = S.launch_profile(T)
    S.set_profiling(T, False)
    print(tag, "launches", len(p["ms"]), "sum ms %.3f" % p["ms"].sum())
    for k in sorted(set(p["kind"].tolist())):
        q = p["kind"] == k
        print("  %-16s n=%4d  ms=%8.3f  GF=%9.2f  TF/s=%6.2f" % (KIND.get(k, k), q.sum(), p["ms"][q].sum(), p["flops"][q].sum() / 1e9,
              p["flops"][q].sum() / 1e12 / max(1e-3 * p["ms"][q].sum(), 1e-30)))
    S.free_factor(Lf); S.free_sparse(A); S.finish()
