#!/usr/bin/env python3
"""What a look-ahead with CU masks could return: per-launch times of one factorization measured
on the whole chip and on k/32 of the CUs of every XCD (tools/launch_profile.py under CHOLMOD_HIP_CU_MASK_32THS), then
per outer block column: today chain + outer update one after the other; with look-ahead the update that completes the
next block column first (U_next), then the rest of the update on (32-k)/32 of the chip beside the chain of the next
block column on k/32.  usage: lookahead_model.py PREFIX  (files PREFIX_cu{0,4,8,16,24,28}.txt)"""
import re
import sys

import numpy as np


def load(path):
    rows = []
    for l in open(path):
        m = re.match(r'\s*(\d+) (\S+)\s+grid=\s*(\d+) aux=\s*(\d+) ms=\s*([\d.]+)', l)
        if m:
            rows.append((int(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4)), float(m.group(5))))
    return rows


pre = sys.argv[1]
T = {k: load(f"{pre}_cu{k}.txt") for k in (0, 4, 8, 16, 24, 28)}
full = T[0]
n = len(full)
ms = {k: np.array([r[4] for r in T[k]]) for k in T}
kind = [r[1] for r in full]
aux = np.array([r[3] for r in full])
grid = np.array([r[2] for r in full])
# an outer update: update_w / update64 with K >= 1024 (the K = OB update that closes an outer block column)
outer = np.array([(kind[i] in ("update_w", "update64") and aux[i] >= 1024) for i in range(n)])
chainkinds = ("potrf", "trsm", "trsm+upd+potrf", "update+potrf", "update64", "update_w")
total = ms[0].sum()
print(f"{pre}: {n} listed launches, {total:.2f} ms on the whole chip; outer updates {int(outer.sum())}, {ms[0][outer].sum():.2f} ms")
idx = np.where(outer)[0]
for k in (4, 8, 16):
    gain = 0.0
    for a, b in zip(idx[:-1], idx[1:]):
        seg = [i for i in range(a + 1, b) if kind[i] in chainkinds]     # the chain between two outer updates
        if not seg or any(kind[i] == "extend_add" for i in range(a + 1, b)):
            continue                                                    # (another level starts in between)
        c_full = ms[0][seg].sum()
        c_part = ms[k][seg].sum()
        u_full = ms[0][a]
        # U_next: the share of the update that lands in the next block column ~ OB / remaining columns; unknown here:
        # take a quarter of the update as a pessimistic figure
        u_next = 0.25 * u_full
        u_rest_part = 0.75 * ms[32 - k if (32 - k) in ms else 28][a] if (32 - k) in ms else 0.75 * u_full * 32.0 / (32 - k)
        now = u_full + c_full
        la = u_next + max(c_part, u_rest_part)
        if la < now:
            gain += now - la
    print(f"  chain on {k}/32 of the CUs: look-ahead returns {gain:.2f} ms of {total:.2f} ({100 * gain / total:.1f} %)")
