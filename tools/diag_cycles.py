#!/usr/bin/env python3
"""Where the time of the 256-column chain's kernels goes: per-phase shader cycles of one k_diag workgroup, its launch
and a k_rowsolve launch in microseconds (cholmod_hip_debug_diag_cycles)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
pr = ch.probes()
names = ["load+left-looking (diag rows)", "T + barrier", "eliminate 64x64", "store + Ls + barrier", "inverses + publish",
         "load+left-looking (row chunks)", "solve row chunks", "closing barrier"]
out = {}
for w, m in ((256, 4096), (128, 4096), (64, 4096), (256, 256)):
    t = np.zeros(10, dtype=np.int64)
    rc = pr.cholmod_hip_debug_diag_cycles(t.ctypes.data, w, m)
    out[f"w{w}_m{m}"] = {"rc": rc, "cycles": {n: int(c) for n, c in zip(names, t[:8])}, "k_diag_us": t[8] / 1e3, "k_rowsolve_us": t[9] / 1e3}
print(json.dumps(out, indent=1))
