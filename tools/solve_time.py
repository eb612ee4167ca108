import time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch, generators as G
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, Ap, Ai, Ax = G.poisson3d(m); perm = G.geometric_nd(m, m, m, 4)
S = ch.Session(factor_on_device=True); A = S.sparse(n, Ap, Ai, Ax, -1); Lf = S.analyze(A, perm); S.factorize(A, Lf)
b = G.demo_rhs(n)
for k in range(3):
    t = time.perf_counter(); x = S.solve(Lf, b); dt = time.perf_counter() - t
    print("solve wall %.1f ms, device %.2f ms" % (dt * 1e3, 1e3 * S.hip_stats(Lf)[24]))
r = G.sym_matvec(n, Ap, Ai, Ax, -1, x) - b; print("resid", np.linalg.norm(r) / np.linalg.norm(b))
