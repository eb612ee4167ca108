#!/usr/bin/env python3
"""Where the one-wave-per-tile update launches of a problem stand against the chip's 2048 tile slots
(256 CUs x 4 SIMDs x 2 waves): tiles per launch, rounds, and what a launch loses when its last round
is partly empty.  Host only (no device).  usage: tail_model.py [grid m] (default 200)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, Ap, Ai, Ax = G.poisson3d(m)
perm = G.geometric_nd(m, m, m, 4)
S = ch.Session(use_gpu=0)
A = S.sparse(n, Ap, Ai, Ax, -1)
Lf = S.analyze(A, perm)
fv = ch.FactorView(Lf)
f = Lf.contents
st = C.c_int(0)
plan = S.L.cholmod_hip_plan_create_dist(fv.n, fv.nsuper, f.super, f.pi, f.px, f.s, ch.HIP_PLAN_HOST_ONLY, 0, 1, C.byref(st))
assert plan and st.value == 0
nl = S.L.cholmod_hip_get_launch_profile(plan, 0, None, None, None, None, None, None)
kind = np.zeros(nl, dtype=np.int32); grid = np.zeros(nl, dtype=np.int32); aux = np.zeros(nl, dtype=np.int32)
ms = np.zeros(nl); fl = np.zeros(nl); by = np.zeros(nl)
S.L.cholmod_hip_get_launch_profile(plan, nl, kind.ctypes.data, grid.ctypes.data, aux.ctypes.data, ms.ctypes.data, fl.ctypes.data, by.ctypes.data)
q = kind == 12
SLOTS = 2048
g, a, F = grid[q].astype(float), aux[q], fl[q]
rounds = g / SLOTS
ideal = F.sum()
quant = (F * np.ceil(rounds) / rounds).sum()
print(f"update_w launches {q.sum()}  flops {F.sum():.3e} of {fl.sum():.3e}")
print(f"quantized-rounds model: time x {quant / ideal:.4f} of ideal")
edges = [0, 1, 2, 4, 8, 16, 32, 64, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    s = (rounds >= lo) & (rounds < hi)
    if s.any():
        print(f"  rounds [{lo:g},{hi:g}): launches {s.sum():4d}  flops {F[s].sum() / ideal * 100:5.1f} %  loss x{(F[s] * np.ceil(rounds[s]) / rounds[s]).sum() / F[s].sum():.3f}  K {a[s].min()}..{a[s].max()}")

# the regions of the launches named on the command line (launch indices of tools/launch_profile.py)
for li in [int(v) for v in sys.argv[2:]]:
    out = np.zeros(12 * 64, dtype=np.int64)
    ng = S.L.cholmod_hip_debug_launch_regions(plan, li, 64, out.ctypes.data)
    print(f"launch {li}: kind {kind[li]} grid {grid[li]} regions {ng}")
    for r in range(min(ng, 64)):
        mm, nn, kk, tri, incb, lda, ldc, ntiles, nblk, front, asg, swz = out[12 * r:12 * r + 12]
        print(f"    m {mm} n {nn} k {kk} tri {tri} cb {incb} lda {lda} ldc {ldc} tiles {ntiles} front {front} assign {asg} swz {swz}")
