import sys, os, json
sys.path.insert(0, '/root/repo')
from suitesparse_amd import cholmod as ch
pr = ch.probes()
TRI, D2, D4, HALF = 65536, 16384, 32768, 524288
out = {}
for k, dep in ((4096, D4), (2048, D4), (1024, D4)):
    for msz in (8192, 10240, 12288, 16384, 24576):
        it = max(1, int(4e11 / (msz * msz * k)))
        tiles = ((msz + 63) // 64) * ((msz + 63) // 64 + 1) // 2
        row = {"tiles": tiles}
        for name, fl in (("update3", TRI | dep), ("update3_half", TRI | dep | HALF)):
            row[name] = pr.cholmod_hip_bench_update_kernel(msz, msz, k, it, fl) / 1e12
        out[f"tri{msz}_K{k}"] = row
for k in (64, 128):
    for (m, n) in ((7000, 64), (7000, 128), (3000, 256), (16000, 128), (1000, 128)):
        row = {"tiles": ((m + 63) // 64) * ((n + 63) // 64)}
        for name, fl in (("update2", 0), ("update3", D2), ("update3_half", D2 | HALF)):
            row[name] = pr.cholmod_hip_bench_update_kernel(m, n, k, max(4, int(5e10 / (m * n * k))), fl) / 1e12
        out[f"rect{m}x{n}_K{k}"] = row
print(json.dumps(out))
