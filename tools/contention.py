#!/usr/bin/env python3
"""k_update3 beside a copy stream (round-5 review, item 8): on 8 GPUs a rank's trailing updates share the memory side with
about 100 GB/s of RCCL traffic in each direction; one wave per tile with no LDS staging draws 2.1 TB/s there already
(12 x its algorithmic bytes, profiles/r05k_pmc_summary_poisson200_top48.json).  Here, on one GPU: the standalone update
kernel (16 384^2 x 4096 and 8192^2 x 512, libcholmod_amd_probes.so) alone, then while a second stream copies device
memory to device memory at a paced rate -- R GB/s of copy is R GB/s read + R GB/s written at the memory side, so 100 GB/s
of copy stands for a reduce-scatter or all-gather at xGMI speed in both directions, 400 for four links' worth.
usage: contention.py [rates_GBps=0,100,200,400,800]"""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from suitesparse_amd import cholmod as ch

rates = [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,100,200,400,800").split(",")]
pr = ch.probes()
pr.cholmod_hip_bench_update_kernel.restype = C.c_double
pr.cholmod_hip_bench_update_kernel.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int]
dev = torch.device("cuda:0")
CH = 64 << 20                        # bytes per copy
src = torch.empty(CH // 8, dtype=torch.float64, device=dev).normal_()
dst = torch.empty_like(src)
side = torch.cuda.Stream(device=dev)
out = []
for shape in ((16384, 16384, 4096, 6), (8192, 8192, 512, 40)):
    m, n, k, iters = shape
    pr.cholmod_hip_bench_update_kernel(m, n, k, 2, 8192)          # warm
    for rate in rates:
        stop = threading.Event()
        moved = [0, 0.0]

        def pump():
            if rate <= 0:
                return
            period = CH / (rate * 1e9)
            t0 = time.perf_counter()
            k_ = 0
            with torch.cuda.stream(side):
                while not stop.is_set():
                    dst.copy_(src, non_blocking=True)
                    k_ += 1
                    # pace on the host clock; never more than 8 copies ahead of the device
                    if k_ % 8 == 0:
                        side.synchronize()
                    lag = t0 + k_ * period - time.perf_counter()
                    if lag > 0:
                        time.sleep(lag)
                side.synchronize()
            moved[0], moved[1] = k_ * CH, time.perf_counter() - t0
        th = threading.Thread(target=pump)
        th.start()
        time.sleep(0.05)
        tf = pr.cholmod_hip_bench_update_kernel(m, n, k, iters, 8192)
        stop.set()
        th.join()
        rec = dict(region=f"{m}x{n}x{k}", copy_GBps_asked=rate, copy_GBps_done=(moved[0] / moved[1] / 1e9) if moved[1] else 0.0,
                   update_TFLOPs=tf / 1e12)
        out.append(rec)
        print(rec, flush=True)
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06_contention.json")
json.dump(out, open(o, "w"), indent=1)
