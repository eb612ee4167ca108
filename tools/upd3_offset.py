#!/usr/bin/env python3
"""k_update3 on the same region with its operands at the start of a small allocation and deep inside a
huge one (CHOLMOD_PROBE_OFFSET_GB): the top fronts of Poisson 200^3 sit 150 GB into the 181.6 GB Lx."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
pr = ch.probes()
out = {}
for gb in ("0", "60", "150", "240"):
    os.environ["CHOLMOD_PROBE_OFFSET_GB"] = gb
    out["offset_%sGB" % gb] = pr.cholmod_hip_bench_update_kernel(49152, 49152, 4096, 3, 65536 | 32768) / 1e12
print(json.dumps(out))
