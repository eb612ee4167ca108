"""Device-memory leak check: 40 analyze / factorize / solve / free cycles, free HBM before and after."""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch, generators as G
L=ch.lib()
n,Ap,Ai,Ax=G.poisson3d(24); perm=G.geometric_nd(24,24,24,4)
def free():
    t=C.c_size_t(0); a=C.c_size_t(0); L.cholmod_hip_memorysize(C.byref(t),C.byref(a)); return a.value
f0=None
for it in range(40):
    S=ch.Session(factor_on_device=bool(it%2)); A=S.sparse(n,Ap,Ai,Ax,-1); Lf=S.analyze(A,perm)
    assert S.factorize(A,Lf)==1
    x=S.solve(Lf,G.demo_rhs(n))
    S.free_factor(Lf); S.free_sparse(A); assert S.cm.malloc_count==0; S.finish()
    if it==4: f0=free()
print('free after 5 iters', f0, 'after 40', free(), 'delta MB', (f0-free())/1e6)
