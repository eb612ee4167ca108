#!/usr/bin/env python3
"""k_update3 (one wave per tile, no LDS) against k_update2: agreement on ragged / triangular /
assign regions, then TFLOP/s on the shapes of the factorization's update launches."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch

pr = ch.probes()
out = {"diff": {}, "TFLOPs": {}}
for fl in (8192, 16384, 32768):
    for (m, n, k, tri, asg) in ((64, 64, 64, 0, 0), (200, 130, 67, 0, 0), (333, 333, 129, 1, 0), (1000, 700, 512, 1, 1),
                                (129, 65, 4, 0, 1), (2049, 2049, 1030, 1, 0), (4096, 4096, 256, 0, 0)):
        out["diff"][f"f{fl}_{m}x{n}x{k}_tri{tri}_asg{asg}"] = pr.cholmod_hip_debug_update_diff(m, n, k, tri, asg, fl)
for (m, n, k, it) in ((16384, 16384, 4096, 2), (16384, 16384, 1024, 4), (8192, 8192, 512, 8), (8192, 8192, 128, 16), (4096, 4096, 64, 32)):
    for name, fl in (("update2_64", 0), ("update3_d3", 8192), ("update3_d2", 16384), ("update3_d4", 32768)):
        out["TFLOPs"][f"{name}_{m}x{n}x{k}"] = pr.cholmod_hip_bench_update_kernel(m, n, k, it, fl) / 1e12
print(json.dumps(out))
