import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
for w in (1, 2, 4):
    print(f"mfma_f64 peak, {w} wave/SIMD: {ch.probes().cholmod_hip_bench_mfma_peak(w, 40000)/1e12:.2f} TFLOP/s", flush=True)
for w in (1, 2, 4, 8):
    print(f"valu fma_f64 peak, {w} wave/SIMD: {ch.probes().cholmod_hip_bench_mfma_peak(-w, 40000)/1e12:.2f} TFLOP/s", flush=True)
