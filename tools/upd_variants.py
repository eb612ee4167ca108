import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
names = {0: "64 bk16 occ2", 256: "64 bk32 occ2", 512: "64 bk16 occ3", 1024: "64 bk16 occ2 dbuf", 4: "128 bk16 occ2",
         2048: "128x64 bk16 occ2", 4096: "64x128 bk16 occ2"}
for (m, n, k) in [(16384, 16384, 512), (16384, 16384, 2048), (16384, 16384, 4096), (4096, 4096, 512)]:
    for fl, nm in names.items():
        print(f"update {m}x{n}x{k} {nm:18s}: {ch.probes().cholmod_hip_bench_update_kernel(m, n, k, 3, fl)/1e12:7.2f} TF", flush=True)
