"""One big launch of the dense update kernel (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
fl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
print(ch.probes().cholmod_hip_bench_update_kernel(m, m, k, 2, fl) / 1e12, "TFLOP/s")
