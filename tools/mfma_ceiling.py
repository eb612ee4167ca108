#!/usr/bin/env python3
"""The fp64 matrix-core ceiling on this box: inline-assembly v_mfma_f64_16x16x4_f64 loop
(cholmod_hip_bench_mfma_ceiling), swept over waves per SIMD x accumulators x operand data.
Prints one JSON line: TFLOP/s by HIP events, shader cycles per MFMA per SIMD, sustained clock."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch


def sweep(pr, seconds=0.3):
    out = {}
    for zero in (0, 1):
        for nacc in (4, 8):
            for w in (1, 2, 4):
                o3 = (C.c_double * 3)()
                # long enough for the clock to settle under the load (DVFS reacts within milliseconds)
                iters = max(int(seconds * 2.3e9 / (4 * nacc * 64 * w)), 64)
                r = pr.cholmod_hip_bench_mfma_ceiling(w, nacc, iters, zero, o3)
                out[f"{'zero' if zero else 'data'}_acc{nacc}_waves{w}"] = {
                    "TFLOPs": r / 1e12, "cycles_per_mfma_per_simd": o3[0], "clock_GHz": o3[1],
                    "TFLOPs_at_2.4GHz_from_issue_rate": o3[2] / 1e12}
    return out


if __name__ == "__main__":
    print(json.dumps(sweep(ch.probes())))
