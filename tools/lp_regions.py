#!/usr/bin/env python3
"""Per-region rates of k_update3 from a tools/launch_profile.py log taken with CHOLMOD_HIP_UPDW_ONE_REGION=1 REGIONS=1 (every
region a launch of its own): shape class (square contribution block / trapezoid of in-front columns), K, tiles, TFLOP/s.
usage: python tools/lp_regions.py log [min_ms]"""
import re
import sys

L = open(sys.argv[1]).read().split("\n")
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
rows = []
for i, l in enumerate(L):
    m = re.match(r'\s*(\d+) update_w\s+grid=\s*(\d+) aux=\s*(\d+) ms=\s*([\d.]+) .*TF/s=\s*([\d.]+)', l)
    if not m or float(m[4]) < min_ms or i + 1 >= len(L) or not L[i + 1].strip().startswith("region"):
        continue
    f = L[i + 1].split()
    r = dict(zip(f[1::2], f[2::2]))
    rows.append((float(m[5]), int(m[1]), int(m[2]), int(m[3]), float(m[4]), int(r["m"]), int(r["n"]), int(r["lda"]), int(r["cb"]), int(r["assign"])))
agg = {}
for tf, idx, grid, k, ms, m_, n_, lda, cb, asg in sorted(rows):
    kind = "cb-square" if cb else ("trapezoid n/m<0.2" if n_ < 0.2 * m_ else "trapezoid" if n_ < m_ else "square")
    a = agg.setdefault(kind, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += tf * ms
    print("%6.2f TF  launch %5d  tiles %7d  K %4d  ms %7.2f  m %6d n %6d lda %6d  %s%s" % (tf, idx, grid, k, ms, m_, n_, lda, kind, " assign" if asg else ""))
for kind, a in agg.items():
    print("%-20s regions %4d  ms %8.1f  TFLOP/s %6.2f" % (kind, a[0], a[1], a[2] / a[1]))
