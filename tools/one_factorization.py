"""Analyze + ONE factorization of a synthetic workload (for rocprofv3 --pmc passes:
no warm-up repeats, no profiling pass, no micro-benchmarks)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch, generators as G
ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=200)
a = ap.parse_args()
m = a.grid
n, Ap, Ai, Ax = G.poisson3d(m)
perm = G.geometric_nd(m, m, m, 4)
S = ch.Session(factor_on_device=True)
A = S.sparse(n, Ap, Ai, Ax, -1)
Lf = S.analyze(A, perm)
t = time.perf_counter()
assert S.factorize(A, Lf) == 1 and S.cm.status == 0
print("one factorization of poisson3d_%d^3: %.2f s (incl. plan + upload)" % (m, time.perf_counter() - t), flush=True)
S.free_factor(Lf); S.free_sparse(A); S.finish()
