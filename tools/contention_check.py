"""Several independent single-rank factorizations sharing one GPU (time-sliced kernels of different processes):
does a factor ever come back wrong?  usage: contention_check.py NPROC REPEATS [grid]   (spawns itself)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "worker":
    rep, m = int(sys.argv[2]), int(sys.argv[3])
    from suitesparse_amd import cholmod as ch
    from suitesparse_amd import generators as G
    from oracle.oracle import OracleFactor
    n, Ap, Ai, Ax = G.poisson3d(m); perm = G.geometric_nd(m, m, m, 4)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    O.factorize(Ax)
    mask = O.lower_mask()
    bad = 0
    for r in range(rep):
        S = ch.Session()
        A = S.sparse(n, Ap, Ai, Ax, -1)
        Lf = S.analyze(A, perm)
        for k in range(2):          # first factorization (search path) and one through the assembly map
            ok = S.factorize(A, Lf)
            fv = ch.FactorView(Lf)
            err = np.linalg.norm((fv.x - O.x)[mask]) / np.linalg.norm(O.x[mask])
            if ok != 1 or S.cm.status != 0 or not (err < 1e-12):
                bad += 1
                # first wrong supernode
                sup, pi, px = fv.super, fv.pi, fv.px
                for s in range(fv.nsuper):
                    a = fv.x[px[s]:px[s + 1]]; b = O.x[px[s]:px[s + 1]]
                    nscol, nsrow = int(sup[s + 1] - sup[s]), int(pi[s + 1] - pi[s])
                    d = np.abs(a - b).reshape(nscol, nsrow).T
                    for j in range(nscol): d[:j, j] = 0
                    if d.max() > 1e-9:
                        print(f"WRONG pid {os.getpid()} rep {r} pass {k}: err {err:.2e} first bad supernode {s} nscol {nscol} nsrow {nsrow} rows {np.where(d.max(axis=1) > 1e-9)[0][[0, -1]]} cols {np.where(d.max(axis=0) > 1e-9)[0][[0, -1]]}", flush=True)
                        break
        S.free_factor(Lf); S.free_sparse(A); S.finish()
    print(f"pid {os.getpid()}: {bad} wrong of {2 * rep}", flush=True)
    sys.exit(1 if bad else 0)

nproc, rep = int(sys.argv[1]), int(sys.argv[2])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(rep), str(m)]) for _ in range(nproc)]
rc = [p.wait() for p in ps]
print("exit codes", rc)
