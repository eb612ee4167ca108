"""Cycles per op of the fp64 instruction patterns the panel kernels chain (one wave)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch

L = ch.lib()
n = 4096
out = np.zeros(8, dtype=np.int64)
print("rc", ch.probes().cholmod_hip_debug_latency(out.ctypes.data, n))
names = ["dependent v_fma_f64", "8 independent v_fma_f64 (per 8)", "dependent v_rcp_f64",
         "8 x (readlane pair + fma) (per 8)", "readlane -> fma dependent", "dependent v_mul_f64",
         "dependent mfma_f64_16x16x4", "LDS read -> fma dependent"]
for nm, c in zip(names, out):
    print(f"{nm:36s} {c / n:8.1f} cycles/rep")
