#!/usr/bin/env python3
"""SHA-256 of the factor a workload produces (L->x after cholmod_l_factorize, downloaded) -- to tell whether a kernel change
is bit-neutral: run it with two builds of the library (CHOLMOD_AMD_LIB=<other .so>) and compare the lines.
usage: python tools/factor_digest.py <workload> <grid>        (bench.py's workloads: poisson3d, poisson2d, box3d)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_workload
from suitesparse_amd import cholmod as ch
w, m = sys.argv[1], int(sys.argv[2])
n, Ap, Ai, Ax, stype, perm, name = build_workload(w, m)
S = ch.Session(ordering="default")
A = S.sparse(n, Ap, Ai, Ax, stype)
Lf = S.analyze(A, perm)
assert S.factorize(A, Lf) == 1
x = ch.FactorView(Lf).x
print("%s  xsize %d  sha256 %s  status %d" % (name, x.size, hashlib.sha256(x.tobytes()).hexdigest()[:32], S.cm.status))
