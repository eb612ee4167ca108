"""Device micro-benchmarks used while tuning (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
for w in (1, 2, 4):
    print(f"mfma_f64 peak, {w} wave/SIMD: {ch.probes().cholmod_hip_bench_mfma_peak(w, 20000)/1e12:.2f} TFLOP/s", flush=True)
names = {0: "v1 128", 8: "v2 128 occ2", 16: "v2 128 occ2 db", 4: "v1 64", 32: "v2 64 occ2"}
for (m, n, k) in [(8192, 8192, 512), (8192, 8192, 64), (16384, 448, 64), (2048, 2048, 512), (1024, 1024, 128), (512, 512, 64)]:
    for fl, nm in names.items():
        r = ch.probes().cholmod_hip_bench_update_kernel(m, n, k, 5, fl)
        print(f"update {m}x{n}x{k} {nm:15s}: {r/1e12:7.2f} TFLOP/s", flush=True)
