#!/usr/bin/env python3
"""What hipMalloc of a factorization-sized reservation costs on this box, and whether it can be had faster: one call of
280 GB / two calls (L, arena) one after the other / the same two from two threads / 8 and 32 pieces from as many threads /
hipMallocAsync from the default pool / the virtual-memory API (hipMemCreate + hipMemMap of 2 GB handles, 8 threads).
Straight on libamdhip64 through ctypes (the calls release the GIL).  usage: python tools/malloc_probe.py [GB=280]"""
import ctypes as C
import json
import sys
import threading
import time

hip = C.CDLL("libamdhip64.so")
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 280.0
TOTAL = int(GB * 1e9) // (1 << 21) * (1 << 21)


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (what, rc))


def malloc(nbytes):
    p = C.c_void_p()
    chk(hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)), "hipMalloc")
    return p


def timed(f):
    t = time.perf_counter()
    r = f()
    return time.perf_counter() - t, r


def pieces(k, threads):
    sizes = [TOTAL // k // (1 << 21) * (1 << 21)] * k
    out = [None] * k

    def work(i):
        out[i] = malloc(sizes[i])
    if threads:
        th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        t, _ = timed(lambda: ([x.start() for x in th], [x.join() for x in th]))
    else:
        t, _ = timed(lambda: [work(i) for i in range(k)])
    tf, _ = timed(lambda: [hip.hipFree(p) for p in out])
    return {"malloc_s": round(t, 3), "free_s": round(tf, 3)}


chk(hip.hipSetDevice(0), "hipSetDevice")
free_b, tot_b = C.c_size_t(), C.c_size_t()
chk(hip.hipMemGetInfo(C.byref(free_b), C.byref(tot_b)), "hipMemGetInfo")
res = {"asked_GB": TOTAL / 1e9, "free_GB": free_b.value / 1e9, "total_GB": tot_b.value / 1e9}
t, p = timed(lambda: malloc(1 << 21))
hip.hipFree(p)
res["first_small_malloc_s"] = round(t, 3)
res["one_call"] = pieces(1, False)
res["one_call_again"] = pieces(1, False)
res["two_calls_in_sequence"] = pieces(2, False)
res["two_threads"] = pieces(2, True)
res["eight_threads"] = pieces(8, True)
res["thirty_two_threads"] = pieces(32, True)
# touching it: a memset of the whole reservation (what the first factorization does to L anyway)
p = malloc(TOTAL)
t, _ = timed(lambda: (hip.hipMemset(p, 0, C.c_size_t(TOTAL)), hip.hipDeviceSynchronize()))
res["memset_all_s"] = round(t, 3)
t, _ = timed(lambda: (hip.hipMemset(p, 0, C.c_size_t(TOTAL)), hip.hipDeviceSynchronize()))
res["memset_all_again_s"] = round(t, 3)
hip.hipFree(p)
# stream-ordered allocation from the default pool
try:
    q = C.c_void_p()
    t, _ = timed(lambda: (chk(hip.hipMallocAsync(C.byref(q), C.c_size_t(TOTAL), None), "hipMallocAsync"), hip.hipDeviceSynchronize()))
    res["malloc_async_s"] = round(t, 3)
    hip.hipFreeAsync(q, None)
    hip.hipDeviceSynchronize()
except Exception as e:          # noqa: BLE001
    res["malloc_async_s"] = repr(e)
print(json.dumps(res))
