#!/usr/bin/env python3
"""Where the API step's time goes: wall time per call of cholmod_l_factorize (values-only path) and of the resident step,
measured around the ctypes calls; CHOLMOD_API_TIMING=1 prints the library's own phases next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_workload
from suitesparse_amd import cholmod as ch
w, m = sys.argv[1], int(sys.argv[2])
if os.environ.get("PROBE_NODE"):        # pin the whole process (all its threads) to one NUMA node's cpus
    txt = open("/sys/devices/system/node/node%s/cpulist" % os.environ["PROBE_NODE"]).read().strip()
    cpus = []
    for part in txt.split(","):
        lo, hi = (int(v) for v in (part.split("-") + [part])[:2])
        cpus += list(range(lo, hi + 1))
    os.sched_setaffinity(0, cpus)
n, Ap, Ai, Ax, stype, perm, name = build_workload(w, m)
S = ch.Session(factor_on_device=True, ordering="default")
A = S.sparse(n, Ap, Ai, Ax, stype)
Lf = S.analyze(A, perm)
assert S.factorize(A, Lf) == 1
for _ in range(3):
    S.refactorize_resident(Lf); S.factorize(A, Lf)
N = 20
t = time.perf_counter()
for _ in range(N):
    S.refactorize_resident(Lf)
tr = (time.perf_counter() - t) / N
t = time.perf_counter()
for _ in range(N):
    S.factorize(A, Lf)
ta = (time.perf_counter() - t) / N
print("%s: resident %.3f ms, api %.3f ms per call (wall around the ctypes call)" % (name, 1e3 * tr, 1e3 * ta))
