#!/usr/bin/env python3
"""Summary of a tools/launch_profile.py log (kind 12 = k_update3) by contraction length K and by launch size.
usage: REPEAT=1 python tools/launch_profile.py poisson3d 200 12 > log; python tools/lp_by_k.py log"""
import collections
import re
import sys

rows = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*(\d+) update_w\s+grid=\s*(\d+) aux=\s*(\d+) ms=\s*([\d.]+) MB=\s*([\d.]+) GB/s=\s*([\d.]+) TF/s=\s*([\d.]+)', l)
    if m:
        rows.append((int(m[2]), int(m[3]), float(m[4]), float(m[7])))
byk = collections.defaultdict(lambda: [0, 0.0, 0.0])
for g, k, ms, tf in rows:
    b = byk[k]; b[0] += 1; b[1] += ms; b[2] += tf * ms
print("by K:      K  launches        ms   TFLOP/s")
for k in sorted(byk):
    b = byk[k]
    print("      %6d  %8d  %8.1f  %8.1f" % (k, b[0], b[1], b[2] / b[1]))
bk = collections.defaultdict(lambda: [0, 0.0, 0.0])
edges = [2048, 10240, 30000, 100000, 300000]
for g, k, ms, tf in rows:
    key = (min(k, 4096), sum(g >= e for e in edges))
    b = bk[key]; b[0] += 1; b[1] += ms; b[2] += tf * ms
print("by K (capped at 4096) and tiles (< 2048, < 10240, < 30000, < 100000, < 300000, more):")
for key in sorted(bk):
    b = bk[key]
    print("      K %5d  size class %d  launches %4d  ms %8.1f  TFLOP/s %6.1f" % (key[0], key[1], b[0], b[1], b[2] / b[1]))
tot = sum(r[2] for r in rows)
print("total ms %.1f, flop-weighted TFLOP/s %.2f" % (tot, sum(r[2] * r[3] for r in rows) / tot))
