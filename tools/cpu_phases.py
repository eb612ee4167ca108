#!/usr/bin/env python3
"""The product's CPU path (Common->useGPU = 0) at one thread count, with its phase timers on (CHOLMOD_CPU_TIMING):
usage: OMP_NUM_THREADS=T python tools/cpu_phases.py [grid=100] [repeats=2]"""
import ctypes, os, sys, time
os.environ.setdefault("CHOLMOD_CPU_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if "CHOLMOD_BLAS_LIBRARY" not in os.environ and bench._scipy_openblas():
    os.environ["CHOLMOD_BLAS_LIBRARY"] = bench._scipy_openblas()
from suitesparse_amd import cholmod as ch, generators as G
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, Ap, Ai, Ax = G.poisson3d(m)
perm = G.geometric_nd(m, m, m, 4)
S = ch.Session(use_gpu=0)
A = S.sparse(n, Ap, Ai, Ax, -1)
Lf = S.analyze(A, perm)
for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    t0 = time.perf_counter()
    assert S.factorize(A, Lf) == 1 and S.cm.status == 0
    dt = time.perf_counter() - t0
    print("threads %s: %.3f s = %.1f GFLOP/s" % (os.environ.get("OMP_NUM_THREADS"), dt, S.cm.fl / dt / 1e9), flush=True)
