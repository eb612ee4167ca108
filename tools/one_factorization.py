"""Analyze + ONE factorization of a synthetic workload (for rocprofv3 --pmc passes:
no warm-up repeats, no profiling pass, no micro-benchmarks).
usage: one_factorization.py [--workload poisson3d|poisson2d|box3d] [--grid M] [--repeat R]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_workload
from suitesparse_amd import cholmod as ch
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="poisson3d")
ap.add_argument("--grid", type=int, default=200)
ap.add_argument("--repeat", type=int, default=0, help="extra factorizations of the resident matrix")
ap.add_argument("--hip-flags", type=int, default=0)
ap.add_argument("--checks", action="store_true",
                help="also run cholmod_hip_factor_checks: k_factor_checks reads Lx exactly once with 8 B/lane "
                     "loads (8*xsize bytes) -- the known-byte kernel the FETCH_SIZE counter is calibrated on")
a = ap.parse_args()
n, Ap, Ai, Ax, stype, perm, name = build_workload(a.workload, a.grid)
S = ch.Session(factor_on_device=True, hip_flags=a.hip_flags)
A = S.sparse(n, Ap, Ai, Ax, stype)
Lf = S.analyze(A, perm)
t = time.perf_counter()
assert S.factorize(A, Lf) == 1 and S.cm.status == 0
print("one factorization of %s: %.2f s (incl. plan + upload)" % (name, time.perf_counter() - t), flush=True)
for _ in range(a.repeat):
    assert S.refactorize_resident(Lf) == 1
if a.checks:
    fv = ch.FactorView(Lf)
    print("factor_checks:", S.factor_checks(Lf), "Lx bytes", 8 * fv.xsize, flush=True)
S.free_factor(Lf); S.free_sparse(A); S.finish()
