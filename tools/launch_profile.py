#!/usr/bin/env python3
"""Per-launch device times of one factorization (HIP events around every launch).
usage: launch_profile.py WORKLOAD M [kind ...]   e.g.  launch_profile.py poisson2d 1259 8"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import build_workload
from suitesparse_amd import cholmod as ch

KIND = {0: "zero", 1: "extend_add", 2: "potrf", 3: "trsm", 4: "update128", 5: "update64", 7: "allreduce", 8: "thin", 9: "update+potrf", 10: "trsm+upd+potrf", 11: "allgather", 12: "update_w", 13: "diag256", 14: "rowsolve", 15: "window", 16: "chain256f"}
w, m = sys.argv[1], int(sys.argv[2])
only = [int(v) for v in sys.argv[3:]]
n, Ap, Ai, Ax, stype, perm, name = build_workload(w, m)
S = ch.Session(factor_on_device=True, hip_flags=int(os.environ.get("HIP_FLAGS", "0")))
A = S.sparse(n, Ap, Ai, Ax, stype)
Lf = S.analyze(A, perm)
assert S.factorize(A, Lf) == 1
S.refactorize_resident(Lf)
S.set_profiling(Lf, True)
best = None
for _ in range(int(os.environ.get("REPEAT", "3"))):
    assert S.refactorize_resident(Lf) == 1
    p = S.launch_profile(Lf)
    best = p if best is None else dict(p, ms=np.minimum(best["ms"], p["ms"]))
p = best
print(name, "launches", len(p["ms"]), "sum ms %.3f" % p["ms"].sum())
for k in sorted(set(p["kind"].tolist())):
    q = p["kind"] == k
    print("  %-10s n=%4d  ms=%8.3f  GB=%8.3f  GF=%9.2f" % (KIND.get(k, k), q.sum(), p["ms"][q].sum(), p["bytes"][q].sum() / 1e9, p["flops"][q].sum() / 1e9))
for i in range(len(p["ms"])):
    if only and p["kind"][i] not in only:
        continue
    if not only and p["ms"][i] < 0.05:
        continue
    gbps = p["bytes"][i] / 1e9 / (p["ms"][i] * 1e-3) if p["ms"][i] > 0 else 0
    tf = p["flops"][i] / 1e12 / (p["ms"][i] * 1e-3) if p["ms"][i] > 0 else 0
    if p["kind"][i] == 8 and os.environ.get("CHOLMOD_HIP_THIN_TIMING"):
        t = np.zeros(10, dtype=np.int64)
        S.L.cholmod_hip_debug_thin_cycles(Lf.contents.hip_plan, i, t.ctypes.data)
        print("      cycles: req+zero %d, A %d, children %d, panels %d, store+barrier %d, trailing %d" % tuple(t[:6]))
    print("%5d %-10s grid=%7d aux=%5d ms=%8.4f MB=%9.2f GB/s=%8.1f TF/s=%6.2f" % (i, KIND.get(p["kind"][i], p["kind"][i]), p["grid"][i], p["aux"][i], p["ms"][i], p["bytes"][i] / 1e6, gbps, tf))
    if os.environ.get("REGIONS") and p["kind"][i] in (4, 5, 9, 12):
        out = np.zeros(12 * 16, dtype=np.int64)
        ng = S.L.cholmod_hip_debug_launch_regions(Lf.contents.hip_plan, i, 16, out.ctypes.data)
        for r in range(min(ng, 16)):
            print("          region m %d n %d k %d tri %d cb %d lda %d ldc %d tiles %d front %d assign %d swz %d" % tuple(out[12 * r + t] for t in (0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11)))
