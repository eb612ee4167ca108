"""Complex (Hermitian) input on a thin-front workload (2D Poisson pattern m^2, geometric ND): complex storage with the
thin-front kernel's complex form against the generic kernels for every front (CHOLMOD_HIP_CX_NO_THIN=1), and the real
factorization of the same pattern.  usage: complex_thin.py [m=800]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np
from suitesparse_amd import cholmod as ch, generators as G
m = int(sys.argv[1]); cx = sys.argv[2] == "1"
n, Ap, Ai, Ax = G.poisson2d(m)
perm = G.geometric_nd(m, m, 1, 4)
vals = G.hermitian_phases(n, Ap, Ai, Ax, seed=m) if cx else Ax
S = ch.Session(factor_on_device=True, ordering="default")
A = S.sparse(n, Ap, Ai, vals, -1)
Lf = S.analyze(A, perm)
assert S.factorize(A, Lf) == 1
for _ in range(2): assert S.refactorize_resident(Lf) == 1
t = time.perf_counter()
for _ in range(10): assert S.refactorize_resident(Lf) == 1
dt = (time.perf_counter() - t) / 10
b = G.demo_rhs(n).astype(vals.dtype)
x = S.solve(Lf, b)
r = (G.herm_matvec(n, Ap, Ai, vals, x) if cx else G.sym_matvec(n, Ap, Ai, vals, -1, x)) - b
print(json.dumps({"ms": 1e3 * dt, "resid": float(np.linalg.norm(r) / np.linalg.norm(b))}))
''' % ROOT
m = sys.argv[1] if len(sys.argv) > 1 else "800"
for tag, cx, env in (("real", "0", {}), ("complex, thin fronts in complex storage", "1", {}), ("complex, generic kernels only", "1", {"CHOLMOD_HIP_CX_NO_THIN": "1"})):
    out = subprocess.run([sys.executable, "-c", CODE, m, cx], env=dict(os.environ, **env), capture_output=True, text=True)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    print(tag, line[-1] if line else out.stderr[-400:], flush=True)
