import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
out = np.zeros(8, dtype=np.int64)
print("rc", ch.probes().cholmod_hip_debug_potrf_cycles(out.ctypes.data))
names = ["stage", "panel update (8 groups)", "publish+barrier", "8x8 factor", "row solve+store", "barrier", "write-back", "-"]
for n, v in zip(names, out):
    print(f"{n:28s} {v:10d} cycles  {v/2.4e3:8.2f} us@2.4GHz")
print("total", out.sum(), out.sum()/2.4e3, "us")
