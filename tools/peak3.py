"""fp64 MFMA issue loop with different fillers between the MFMAs (see k_mfma_peak)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
for fill, nm in ((0, "back to back"), (1, "s_nop 3"), (2, "v_fma_f32"), (3, "ds_read_b64")):
    for w in (1, 2, 4):
        r = ch.probes().cholmod_hip_bench_mfma_peak(1000 * fill + w, 20000)
        print(f"fill {nm:14s} {w} wave(s)/SIMD: {r / 1e12:6.2f} TFLOP/s", flush=True)
