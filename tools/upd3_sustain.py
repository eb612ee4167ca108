#!/usr/bin/env python3
"""Is the update kernel's rate inside a long factorization a sustained-power effect?  The same launch
(k_update3, triangular 49 152^2 region, K = 4096) repeated for ~0.7 s, ~4 s and ~12 s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
pr = ch.probes()
out = {}
for it in (2, 12, 36, 2):
    out[f"u3_tri48k_K4096_iters{it}" + ("_again" if f"u3_tri48k_K4096_iters{it}" in out else "")] = \
        pr.cholmod_hip_bench_update_kernel(49152, 49152, 4096, it, 65536 | 32768) / 1e12
print(json.dumps(out))
