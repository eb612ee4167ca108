#!/usr/bin/env python3
"""Which side of k_update3 pays for an unaligned front: the operands (lda, 8-byte aligned columns of the panel) or the target
(ldc, the read-modify-write of C)?  Triangular regions of the size of Poisson 200^3's fronts, K = 4096 / 1024, four layouts.
usage: python tools/ld_parts.py      prints one JSON line"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json
sys.path.insert(0, %r)
from suitesparse_amd import cholmod as ch
pr = ch.probes()
TRI, ODD, D4 = 65536, 131072, 32768
o = {}
for name, (m, n, k, it) in {"tri40k_K4096": (40000, 40000, 4096, 1), "tri24k_K4096": (24000, 24000, 4096, 2), "tri24k_K1024": (24000, 24000, 1024, 4),
                            "trap48kx8k_K4096": (48000, 8000, 4096, 2)}.items():
    o[name] = pr.cholmod_hip_bench_update_kernel(m, n, k, it, TRI | D4 | int(sys.argv[1])) / 1e12
print(json.dumps(o))
""" % ROOT
out = {}
for label, flag, parts in (("aligned", 0, "3"), ("operands_odd", 131072, "1"), ("target_odd", 131072, "2"), ("both_odd", 131072, "3")):
    env = dict(os.environ, CHOLMOD_PROBE_ODD_PARTS=parts)
    r = subprocess.run([sys.executable, "-c", CHILD, str(flag)], env=env, capture_output=True, text=True, timeout=280)
    out[label] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else {"error": r.stderr[-300:]}
print(json.dumps(out))
