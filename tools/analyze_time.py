"""Host-side timing of analyze + plan creation (no factorization).
usage: analyze_time.py GRID   (env: OMP_NUM_THREADS, CHOLMOD_ANALYZE_TIMING, CHOLMOD_HIP_PLAN_TIMING)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch, generators as G
m = int(sys.argv[1])
n, Ap, Ai, Ax = G.poisson3d(m)
perm = G.geometric_nd(m, m, m, 4)
S = ch.Session(factor_on_device=True)
A = S.sparse(n, Ap, Ai, Ax, -1)
for rep in range(2):
    t = time.perf_counter(); Lf = S.analyze(A, perm); ta = time.perf_counter() - t
    t = time.perf_counter(); ok = S.L.cholmod_l_hip_prepare(Lf, C.byref(S.cm)); tp = time.perf_counter() - t
    print("threads %s grid %d: analyze %.3f s, plan %.3f s (ok %d)" % (os.environ.get("OMP_NUM_THREADS", "default"), m, ta, tp, ok), flush=True)
    S.free_factor(Lf)
