import sys, os, json
sys.path.insert(0, os.getcwd())
from suitesparse_amd import cholmod as ch
pr = ch.probes()
TRI, NOSWZ, ODD, D4 = 65536, 32, 131072, 32768
out = {}
for name, (m, n, k, it, fl) in {
        "ld50888_n2896": (50888, 2896, 4096, 2, TRI | D4), "ld50889_n2896": (50889, 2896, 4096, 2, TRI | D4),
        "ld50880_n2880": (50880, 2880, 4096, 2, TRI | D4), "ld50888_n2896_odd": (50888, 2896, 4096, 2, TRI | D4 | ODD),
        "ld50888_n6992": (50888, 6992, 4096, 1, TRI | D4), "ld50944_n2896": (50944, 2896, 4096, 2, TRI | D4),
        "ld50888_n2880": (50888, 2880, 4096, 2, TRI | D4), "ld50880_n2896": (50880, 2896, 4096, 2, TRI | D4),
        "ld25452_n1655": (25452, 1655, 4096, 4, TRI | D4), "ld25472_n1664": (25472, 1664, 4096, 4, TRI | D4),
        }.items():
    out[name] = round(pr.cholmod_hip_bench_update_kernel(m, n, k, it, fl) / 1e12, 2)
print(json.dumps(out))
