#!/usr/bin/env python3
"""The update kernels on regions of the size the top fronts of Poisson 200^3 have (a triangular
49 152^2 region, K = 4096: 1.6 GB of operand panel, 19 GB of target), against the 16 384^2 square
the round-3 tuning used; with and without the XCD-aware tile walk; and the MFMA ceiling over 3 s."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch

pr = ch.probes()
TRI, NOSWZ, ODD = 65536, 32, 131072          # NOSWZ = CHOLMOD_HIP_NO_XCD_SWIZZLE
out = {}
for name, (m, n, k, it, fl) in {
        "u2_sq16k_K4096": (16384, 16384, 4096, 2, 0), "u3_sq16k_K4096": (16384, 16384, 4096, 2, 32768),
        "u2_tri48k_K4096": (49152, 49152, 4096, 1, TRI), "u3_tri48k_K4096": (49152, 49152, 4096, 1, TRI | 32768),
        "u3_tri48k_K4096_noswz": (49152, 49152, 4096, 1, TRI | 32768 | NOSWZ),
        "u3_tri48k_K1024": (49152, 49152, 1024, 2, TRI | 32768), "u3_tri48k_K512": (49152, 49152, 512, 2, TRI | 16384),
        "u3_tri24k_K4096": (24576, 24576, 4096, 2, TRI | 32768),
        "u3_tri48k_K4096_oddlayout": (49153, 49153, 4096, 1, TRI | 32768 | ODD), "u2_tri48k_K4096_oddlayout": (49153, 49153, 4096, 1, TRI | ODD),
        "u3_tri48k_K512_oddlayout": (49153, 49153, 512, 2, TRI | 16384 | ODD),
        "u3_tri12k_K4096": (12288, 12288, 4096, 4, TRI | 32768), "u3_tri6k_K2048": (6144, 6144, 2048, 8, TRI | 32768),
        "u2_tri6k_K2048": (6144, 6144, 2048, 8, TRI),
}.items():
    out[name] = pr.cholmod_hip_bench_update_kernel(m, n, k, it, fl) / 1e12
    print(name, out[name], file=sys.stderr, flush=True)
o3 = (C.c_double * 3)()
r = pr.cholmod_hip_bench_mfma_ceiling(1, 8, int(3.0 * 2.3e9 / (4 * 8 * 64)), 0, o3)
out["ceiling_3s_data_acc8_waves1"] = {"TFLOPs": r / 1e12, "clock_GHz": o3[1]}
print(json.dumps(out))
