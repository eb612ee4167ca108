#!/usr/bin/env python3
"""CU masks on this part (tuning; DESIGN section 4, look-ahead): where the workgroups of a stream created with
hipExtStreamCreateWithCUMask land, and what the one-wave-per-tile update and a chain of small dependent launches
cost each other on two masked streams.
usage: cumask.py where | overlap"""
import ctypes as C
import json
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from suitesparse_amd import cholmod as ch

pr = ch.probes()
mode = sys.argv[1] if len(sys.argv) > 1 else "where"


def words(bits):
    w = np.zeros(8, dtype=np.uint32)
    for b in bits:
        w[b // 32] |= np.uint32(1 << (b % 32))
    return w


def where(mask, blocks=16384, spin=20):
    out = np.zeros(blocks, dtype=np.int64)
    ms = pr.cholmod_hip_probe_cu_mask(mask.ctypes.data if mask is not None else None, 0 if mask is None else len(mask), blocks, spin, out.ctypes.data)
    xcc = (out >> 32) & 0xF
    hw = out & 0xFFFFFFFF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 0x7
    places = Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = Counter(xcc.tolist())
    return {"ms": ms, "distinct_cus": len(places), "cus_per_xcc": {int(k): len([p for p in places if p[0] == k]) for k in sorted(per_xcc)},
            "workgroups_per_xcc": {int(k): int(v) for k, v in sorted(per_xcc.items())}}


res = {}
if mode == "where":
    res["no_mask"] = where(None)
    res["all_256_bits"] = where(words(range(256)))
    res["bits_0_31"] = where(words(range(32)))
    res["bits_0_127"] = where(words(range(128)))
    res["bits_32_255"] = where(words(range(32, 256)))
    res["even_bits"] = where(words(range(0, 256, 2)))
    res["bits_mod8_eq_0"] = where(words(range(0, 256, 8)))
    res["low4_of_each_word"] = where(words([32 * w + b for w in range(8) for b in range(4)]))
    res["all_but_low4_of_each_word"] = where(words([32 * w + b for w in range(8) for b in range(4, 32)]))
elif mode == "overlap":
    o = np.zeros(4)
    cases = {"no_masks": (None, None),
             "7_of_8_words_vs_1": (words(range(32, 256)), words(range(32))),
             "28_of_32_per_word_vs_4": (words([32 * w + b for w in range(8) for b in range(4, 32)]), words([32 * w + b for w in range(8) for b in range(4)]))}
    for name, (ma, mb) in cases.items():
        for (m, k) in ((16384, 1024), (24576, 4096)):
            rc = pr.cholmod_hip_probe_overlap(ma.ctypes.data if ma is not None else None, mb.ctypes.data if mb is not None else None,
                                              0 if ma is None else 8, m, k, 40, 64, 20, o.ctypes.data)
            res[f"{name}_tri{m}_K{k}"] = {"rc": rc, "update_alone_ms": o[0], "chain_alone_ms": o[1], "together_update_ms": o[2], "together_chain_ms": o[3]}
print(json.dumps(res, indent=1))
