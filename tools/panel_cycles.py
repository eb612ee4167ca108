"""Per-phase shader-clock cycles of the matrix-core panel kernels (one workgroup)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch

L = ch.lib()
out = np.zeros(16, dtype=np.int64)
print("rc", ch.probes().cholmod_hip_debug_panel_cycles(out.ctypes.data))
pn = ["stage", "column chain (4 panels)", "scale+store", "barrier", "trailing MFMA tiles", "barrier", "write-back", "-"]
tn = ["stage L11/B", "reciprocals", "diag inverses", "update MFMAs", "barrier", "diag MFMAs+store", "barrier", "-"]
for title, names, v in (("k_potrf_mfma", pn, out[:8]), ("k_trsm_mfma", tn, out[8:])):
    print(title)
    for n, c in zip(names[:7], v[:7]):
        print(f"  {n:26s} {c:9d} cycles {c / 2.4e3:7.2f} us@2.4GHz")
    print(f"  total {v[:7].sum()} cycles {v[:7].sum() / 2.4e3:.2f} us")
    print(f"  whole launch by HIP events: {v[7] / 1e3:.2f} us" + ("  (one workgroup)" if title.startswith("k_potrf") else "  (16 000 rows, 250 workgroups)"))
