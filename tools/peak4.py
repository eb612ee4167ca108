"""MFMA issue loops with the update kernel's operand pattern (accumulator grid per wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
for var in (110, 220, 221, 240, 241, 420, 421, 440, 441, 140, 141):
    row = []
    for w in (1, 2, 3, 4, 6, 8):
        r = ch.probes().cholmod_hip_bench_mfma_peak2(var, w, 40000 // (w * max(1, (var // 100) * ((var // 10) % 10) // 4)), 0)
        row.append("%5.1f" % (r / 1e12))
    print("ti=%d tj=%d lds=%d  waves 1,2,3,4,6,8: %s" % (var // 100, (var // 10) % 10, var % 10, " ".join(row)), flush=True)
