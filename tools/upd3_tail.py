#!/usr/bin/env python3
"""k_update3 with contraction lengths that are no multiple of 4 (the last outer block of a front): agreement
with k_update2 and the rate next to a multiple of 4."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
pr = ch.probes()
out = {"diff": {}, "TFLOPs": {}}
for fl in (16384, 32768):
    for (m, n, k, tri, asg) in ((64, 64, 1, 0, 0), (200, 130, 67, 0, 0), (333, 333, 129, 1, 0), (1000, 700, 513, 1, 1), (129, 65, 3, 0, 1),
                                (2049, 2049, 1030, 1, 0), (512, 512, 7, 0, 0), (640, 640, 9, 1, 0), (640, 640, 10, 1, 0), (640, 640, 11, 1, 0),
                                (640, 640, 17, 0, 0), (640, 640, 23, 0, 0)):
        out["diff"][f"f{fl}_{m}x{n}x{k}_tri{tri}_asg{asg}"] = pr.cholmod_hip_debug_update_diff(m, n, k, tri, asg, fl)
for k in (3432, 3433, 1349, 1348):
    out["TFLOPs"][f"u3_tri24k_K{k}"] = pr.cholmod_hip_bench_update_kernel(24576, 24576, k, 2, 65536 | 32768) / 1e12
print(json.dumps(out))
