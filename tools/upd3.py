#!/usr/bin/env python3
"""k_update3 (one wave per 64 x 64 tile, no LDS) against k_update2, one mode per question asked in round 3:
  standalone  agreement on ragged / triangular / assign regions, TFLOP/s on the shapes of the update launches
  big         regions of the size of Poisson 200^3's top fronts (triangular 49 152^2, K = 4096), XCD walk on / off,
              a real front's odd layout, the MFMA ceiling over 3 s
  tail        contraction lengths that are no multiple of 4: agreement and rate
  sustain     the same launch for 0.3 / 1.6 / 4.8 s (is the rate inside a long factorization a power effect?)
  offset      the operands 0 / 60 / 150 / 240 GB into one allocation (does it matter where a front lives?)
  rounds      triangular regions of 6k .. 48k rows at K = 4096 / 1024: rate against the number of rounds of 2048 tiles
  trap        tall trapezoids (the in-front part of a top front's outer update) against the square of its contribution block
  pair        a top front's outer update (trapezoid + square on one panel) as one launch, as two, and re-ordered
usage: python tools/upd3.py [mode]      prints one JSON line"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch

pr = ch.probes()
TRI, NOSWZ, ODD, D2, D3, D4, SWZ16 = 65536, 32, 131072, 16384, 8192, 32768, 262144
mode = sys.argv[1] if len(sys.argv) > 1 else "standalone"
out = {}
if mode == "standalone":
    out = {"diff": {}, "TFLOPs": {}}
    for fl in (D3, D2, D4):
        for (m, n, k, tri, asg) in ((64, 64, 64, 0, 0), (200, 130, 67, 0, 0), (333, 333, 129, 1, 0), (1000, 700, 512, 1, 1),
                                    (129, 65, 4, 0, 1), (2049, 2049, 1030, 1, 0), (4096, 4096, 256, 0, 0)):
            out["diff"][f"f{fl}_{m}x{n}x{k}_tri{tri}_asg{asg}"] = pr.cholmod_hip_debug_update_diff(m, n, k, tri, asg, fl)
    for (m, n, k, it) in ((16384, 16384, 4096, 2), (16384, 16384, 1024, 4), (8192, 8192, 512, 8), (8192, 8192, 128, 16), (4096, 4096, 64, 32)):
        for name, fl in (("update2_64", 0), ("update3_d3", D3), ("update3_d2", D2), ("update3_d4", D4)):
            out["TFLOPs"][f"{name}_{m}x{n}x{k}"] = pr.cholmod_hip_bench_update_kernel(m, n, k, it, fl) / 1e12
elif mode == "big":
    for name, (m, n, k, it, fl) in {
            "u2_sq16k_K4096": (16384, 16384, 4096, 2, 0), "u3_sq16k_K4096": (16384, 16384, 4096, 2, D4),
            "u2_tri48k_K4096": (49152, 49152, 4096, 1, TRI), "u3_tri48k_K4096": (49152, 49152, 4096, 1, TRI | D4),
            "u3_tri48k_K4096_noswz": (49152, 49152, 4096, 1, TRI | D4 | NOSWZ),
            "u3_tri48k_K1024": (49152, 49152, 1024, 2, TRI | D4), "u3_tri48k_K512": (49152, 49152, 512, 2, TRI | D2),
            "u3_tri24k_K4096": (24576, 24576, 4096, 2, TRI | D4),
            "u3_tri48k_K4096_oddlayout": (49153, 49153, 4096, 1, TRI | D4 | ODD), "u2_tri48k_K4096_oddlayout": (49153, 49153, 4096, 1, TRI | ODD),
            "u3_tri48k_K512_oddlayout": (49153, 49153, 512, 2, TRI | D2 | ODD),
            "u3_tri12k_K4096": (12288, 12288, 4096, 4, TRI | D4), "u3_tri6k_K2048": (6144, 6144, 2048, 8, TRI | D4),
            "u2_tri6k_K2048": (6144, 6144, 2048, 8, TRI)}.items():
        out[name] = pr.cholmod_hip_bench_update_kernel(m, n, k, it, fl) / 1e12
    o3 = (C.c_double * 3)()
    r = pr.cholmod_hip_bench_mfma_ceiling(1, 8, int(3.0 * 2.3e9 / (4 * 8 * 64)), 0, o3)
    out["ceiling_3s_data_acc8_waves1"] = {"TFLOPs": r / 1e12, "clock_GHz": o3[1]}
elif mode == "tail":
    out = {"diff": {}, "TFLOPs": {}}
    for fl in (D2, D4):
        for (m, n, k, tri, asg) in ((64, 64, 1, 0, 0), (200, 130, 67, 0, 0), (333, 333, 129, 1, 0), (1000, 700, 513, 1, 1), (129, 65, 3, 0, 1),
                                    (2049, 2049, 1030, 1, 0), (512, 512, 7, 0, 0), (640, 640, 9, 1, 0), (640, 640, 10, 1, 0), (640, 640, 11, 1, 0),
                                    (640, 640, 17, 0, 0), (640, 640, 23, 0, 0)):
            out["diff"][f"f{fl}_{m}x{n}x{k}_tri{tri}_asg{asg}"] = pr.cholmod_hip_debug_update_diff(m, n, k, tri, asg, fl)
    for k in (3432, 3433, 1349, 1348):
        out["TFLOPs"][f"u3_tri24k_K{k}"] = pr.cholmod_hip_bench_update_kernel(24576, 24576, k, 2, TRI | D4) / 1e12
elif mode == "half":
    # (round 5) half tiles -- two waves per 64 x 64 tile -- on the region sizes of the mid-size problems' top fronts:
    # agreement with k_update2 on ragged / triangular / assign regions, then rate against the whole-tile form and k_update2
    HALF = 524288
    out = {"diff": {}, "TFLOPs": {}}
    for fl in (HALF | D2, HALF | D4):
        for (m, n, k, tri, asg) in ((64, 64, 64, 0, 0), (200, 130, 67, 0, 0), (333, 333, 129, 1, 0), (1000, 700, 512, 1, 1), (129, 65, 4, 0, 1),
                                    (129, 33, 5, 0, 0), (129, 31, 7, 0, 1), (2049, 2049, 1030, 1, 0), (1056, 1056, 260, 1, 0), (4096, 4096, 256, 0, 0)):
            out["diff"][f"f{fl}_{m}x{n}x{k}_tri{tri}_asg{asg}"] = pr.cholmod_hip_debug_update_diff(m, n, k, tri, asg, fl)
    for k, dep in ((1024, D4), (512, D2), (256, D2)):
        for msz in (7560, 6536, 5512, 4488, 3464, 2440, 1416):
            it = max(2, int(2e11 / (msz * msz * k)))
            tiles = ((msz + 63) // 64) * ((msz + 63) // 64 + 1) // 2
            row = {"tiles": tiles}
            for name, fl in (("update2", TRI), ("update3", TRI | dep), ("update3_half", TRI | dep | HALF)):
                row[name] = pr.cholmod_hip_bench_update_kernel(msz, msz, k, it, fl) / 1e12
            out["TFLOPs"][f"tri{msz}_K{k}"] = row
    for (m, n, k) in ((7000, 1024, 1024), (7000, 512, 512), (4000, 2048, 512), (12000, 2048, 2048)):
        row = {"tiles": ((m + 63) // 64) * ((n + 63) // 64)}
        dep = D4 if k >= 1024 else D2
        for name, fl in (("update2", 0), ("update3", dep), ("update3_half", dep | HALF)):
            row[name] = pr.cholmod_hip_bench_update_kernel(m, n, k, max(2, int(2e11 / (m * n * k))), fl) / 1e12
        out["TFLOPs"][f"rect{m}x{n}_K{k}"] = row
elif mode == "trap":
    # (round 5) the outer update of a top front = a tall trapezoid (the in-front columns right of the outer block) + the square of
    # the contribution block: in the 200^3 factorization the launches with a NARROW trapezoid run at 66-69 TFLOP/s, the launches
    # of one square region at 74-75 (tools/lp_by_k.py).  The trapezoid alone, as a rectangle, and the square, same K:
    for name, (m, n, k, it, fl) in {
            "trap_42397x2797": (42397, 2797, 4096, 2, TRI | D4 | ODD), "rect_42397x2797": (42397, 2797, 4096, 2, D4 | ODD),
            "trap_46493x6893": (46493, 6893, 4096, 1, TRI | D4 | ODD), "trap_58705x18705": (58705, 18705, 4096, 1, TRI | D4 | ODD),
            "tri_39600": (39600, 39600, 4096, 1, TRI | D4 | ODD), "trap_42397x2797_noswz": (42397, 2797, 4096, 2, TRI | D4 | ODD | NOSWZ),
            "trap_42397x2797_swz16": (42397, 2797, 4096, 2, TRI | D4 | ODD | SWZ16),
            "trap_42397x2797_even": (42397, 2797, 4096, 2, TRI | D4), "trap_42368x2816_even": (42368, 2816, 4096, 2, TRI | D4),
            "trap_42397x512": (42397, 512, 4096, 4, TRI | D4 | ODD), "trap_42397x1024": (42397, 1024, 4096, 4, TRI | D4 | ODD)}.items():
        out[name] = pr.cholmod_hip_bench_update_kernel(m, n, k, it, fl) / 1e12
elif mode == "pair":
    # (round 5) the outer update of a top front of the 200^3 factorization -- trapezoid 42696 x 2896 and square 39800^2 on one
    # panel of ld 50888, K = 4096 -- as ONE launch (the engine before the split: 67 TFLOP/s there), as two launches (what ships),
    # as one launch with the square first, and with the square's blocks starting at a multiple of 16 384
    for name, (m1, n1, m2, k, ld) in {"front_50888_ob1": (42696, 2896, 39800, 4096, 50888), "front_50888_ob0": (46792, 6992, 39800, 4096, 50888),
                                      "front_62801_ob4": (42321, 2321, 40000, 4096, 62801), "mid_100cubed": (11000, 1024, 9900, 4096, 14000)}.items():
        out[name] = {mname: pr.cholmod_hip_bench_update_pair(m1, n1, m2, k, ld, 2, md) / 1e12
                     for mname, md in (("one_launch", 0), ("two_launches", 1), ("one_launch_square_first", 2), ("one_launch_square_at_16384", 3))}
elif mode == "sustain":
    for it in (2, 12, 36):
        out[f"u3_tri48k_K4096_iters{it}"] = pr.cholmod_hip_bench_update_kernel(49152, 49152, 4096, it, TRI | D4) / 1e12
elif mode == "offset":
    for gb in ("0", "60", "150", "240"):
        os.environ["CHOLMOD_PROBE_OFFSET_GB"] = gb
        out["offset_%sGB" % gb] = pr.cholmod_hip_bench_update_kernel(49152, 49152, 4096, 3, TRI | D4) / 1e12
elif mode == "swz16diff":
    out = {f"{m}x{n}x{k}_tri{tri}_asg{asg}": pr.cholmod_hip_debug_update_diff(m, n, k, tri, asg, D4 | SWZ16)
           for (m, n, k, tri, asg) in ((2049, 2049, 130, 1, 0), (4096, 4096, 64, 0, 0), (5000, 3000, 67, 1, 1), (3333, 3333, 129, 1, 0), (64, 64, 8, 0, 0), (1100, 900, 40, 0, 1))}
elif mode == "swz16":
    # 16 x 16 super-tiles per XCD against 8 x 8 (speed; the fetch side: rocprofv3 --pmc FETCH_SIZE over this mode)
    for name, (m, n, k, it, fl) in {
            "u3_tri48k_K4096_8x8": (49152, 49152, 4096, 1, TRI | D4), "u3_tri48k_K4096_16x16": (49152, 49152, 4096, 1, TRI | D4 | SWZ16),
            "u3_tri24k_K4096_8x8": (24576, 24576, 4096, 2, TRI | D4), "u3_tri24k_K4096_16x16": (24576, 24576, 4096, 2, TRI | D4 | SWZ16),
            "u3_tri24k_K1024_8x8": (24576, 24576, 1024, 4, TRI | D4), "u3_tri24k_K1024_16x16": (24576, 24576, 1024, 4, TRI | D4 | SWZ16),
            "u3_sq16k_K4096_8x8": (16384, 16384, 4096, 2, D4), "u3_sq16k_K4096_16x16": (16384, 16384, 4096, 2, D4 | SWZ16)}.items():
        out[name] = pr.cholmod_hip_bench_update_kernel(m, n, k, it, fl) / 1e12
elif mode == "fetch":
    # under rocprofv3 --pmc FETCH_SIZE: dispatches of k_update3<4> in order = [8x8 warm-up, 8x8, 16x16 warm-up, 16x16] x shapes
    for name, (m, n, k, it, fl) in {
            "u3_tri48k_K4096_8x8": (49152, 49152, 4096, 1, TRI | D4), "u3_tri48k_K4096_16x16": (49152, 49152, 4096, 1, TRI | D4 | SWZ16),
            "u3_tri24k_K1024_8x8": (24576, 24576, 1024, 1, TRI | D4), "u3_tri24k_K1024_16x16": (24576, 24576, 1024, 1, TRI | D4 | SWZ16)}.items():
        out[name] = pr.cholmod_hip_bench_update_kernel(m, n, k, it, fl) / 1e12
elif mode == "rounds":
    for k in (4096, 1024):
        for nrows in (6144, 8192, 12288, 16384, 20480, 24576, 28672, 32768, 40960, 49152):
            nt = nrows // 64
            tiles = nt * (nt + 1) // 2
            tf = pr.cholmod_hip_bench_update_kernel(nrows, nrows, k, 2, TRI | D4) / 1e12
            out[f"tri{nrows}_K{k}"] = {"tiles": tiles, "rounds": tiles / 2048.0, "TFLOPs": tf,
                                       "ms_per_launch": 2.0 * (nrows * (nrows + 1) / 2) * k / tf / 1e9}
else:
    raise SystemExit(__doc__)
print(json.dumps(out))
