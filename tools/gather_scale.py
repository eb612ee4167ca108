import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ctypes as C
import torch
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G
from suitesparse_amd.dist import _DevView
m, W, r = 200, 2, 0
n, Ap, Ai, Ax = G.poisson3d(m); perm = G.geometric_nd(m, m, m, 4)
def _fn(ptr, count, first, size, user):
    if count in (2 * W, 3 * W):
        t = torch.as_tensor(_DevView(ptr, count), device="cuda"); t[:W] = 1e18; torch.cuda.synchronize()
    return 0
cb = ch.ALLREDUCE_FN(_fn)
S = ch.Session(factor_on_device=True, rank=r, world=W, allreduce=cb)
A = S.sparse(n, Ap, Ai, Ax, -1)
Lf = S.analyze(A, perm)
assert S.factorize(A, Lf) == 1
st = S.hip_stats(Lf)
free0, tot = torch.cuda.mem_get_info()
t0 = time.perf_counter()
g = S.L.cholmod_l_gather_factor(Lf, C.byref(S.cm))
t1 = time.perf_counter()
free1, _ = torch.cuda.mem_get_info()
print(json.dumps({"gather_rc": int(g), "status": int(S.cm.status), "seconds": t1 - t0, "L_own_GB": st[36] / 1e9, "L_whole_GB": st[5] / 1e9,
                  "free_before_GB": free0 / 1e9, "free_after_GB": free1 / 1e9, "mem_available_GB_host": [float(l.split()[1]) / 1e6 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]}))
t0 = time.perf_counter()
ok = S.refactorize_resident(Lf)
print(json.dumps({"refactorize_after_staged_gather": int(ok), "seconds": time.perf_counter() - t0}))
