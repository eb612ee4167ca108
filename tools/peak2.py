import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
for w in (1, 2, 3, 4, 8):
    print(f"mfma_f64 8 acc, {w} wave/SIMD: {ch.probes().cholmod_hip_bench_mfma_peak(w, 20000)/1e12:.2f} TF   16 acc: {ch.probes().cholmod_hip_bench_mfma_peak(100 + w, 10000)/1e12:.2f} TF", flush=True)
for (m, n, k) in [(8192, 8192, 512), (16384, 16384, 512), (32768, 32768, 512), (16384, 16384, 2048)]:
    print(f"update {m}x{n}x{k}: {ch.probes().cholmod_hip_bench_update_kernel(m, n, k, 3, 0)/1e12:.2f} TF", flush=True)
