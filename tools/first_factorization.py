#!/usr/bin/env python3
"""What the FIRST cholmod_l_factorize of a symbolic factor costs beside a steady step (round-5 review, item 6): analyze, plan
build + upload (cholmod_l_hip_prepare, or inside analyze when Common->useGPU is on), the long way of the first factorization
(permutation, upload of S, search-based assembly, relative maps) -- the library's own phase timers are switched on
(CHOLMOD_API_TIMING, CHOLMOD_HIP_PLAN_TIMING, CHOLMOD_HIP_HOST_TIMING).  usage: first_factorization.py [grid=200]"""
import ctypes as C, os, sys, time
os.environ.setdefault("CHOLMOD_API_TIMING", "1")
os.environ.setdefault("CHOLMOD_HIP_PLAN_TIMING", "1")
os.environ.setdefault("CHOLMOD_HIP_HOST_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_workload
from suitesparse_amd import cholmod as ch
m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.perf_counter()
n, Ap, Ai, Ax, stype, perm, name = build_workload("poisson3d", m)
t1 = time.perf_counter()
S = ch.Session(factor_on_device=True, ordering="default")
A = S.sparse(n, Ap, Ai, Ax, stype)
t2 = time.perf_counter()
Lf = S.analyze(A, perm)
t3 = time.perf_counter()
have_plan = bool(Lf.contents.hip_plan)
ok = S.L.cholmod_l_hip_prepare(Lf, C.byref(S.cm))
t4 = time.perf_counter()
assert ok == 1, S.cm.status
assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
t5 = time.perf_counter()
assert S.factorize(A, Lf) == 1
t6 = time.perf_counter()
assert S.refactorize_resident(Lf) == 1
t7 = time.perf_counter()
print("%s: generate %.2f s, copy in %.2f, analyze %.2f (plan inside: %s), prepare %.2f, FIRST factorize %.2f, second (values only) %.2f, resident %.2f"
      % (name, t1 - t0, t2 - t1, t3 - t2, have_plan, t4 - t3, t5 - t4, t6 - t5, t7 - t6))
print("plan creation %.2f s (%s); everything between the end of the analysis and a numeric factor: %.2f s; analyze + that: %.2f s"
      % (S.cm.hip_plan_seconds, "inside cholmod_l_analyze" if have_plan else "cholmod_l_hip_prepare", t5 - t3, t5 - t2))
