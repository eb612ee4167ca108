#!/usr/bin/env python3
"""Which XCD does block b of a one-wave-per-workgroup launch run on?  k_update3's tile walk (kernels.hip.h: decode_tile) assumes
blockIdx % 8, for every block of a launch of several hundred thousand.  Launches `blocks` workgroups that spin for spin_us
(negative: uneven durations, 1 .. 4 times as long by a hash of the block index) and reports how many ran on XCD != b % 8, by
position in the launch.   usage: python tools/xcd_map.py [blocks=300000] [spin_us=-20]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch

pr = ch.probes()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
spin = int(sys.argv[2]) if len(sys.argv) > 2 else -20
out = np.zeros(blocks, dtype=np.int64)
ms = pr.cholmod_hip_probe_cu_mask(None, 0, blocks, spin, out.ctypes.data)
xcc = (out >> 32) & 0xF
b = np.arange(blocks)
mis = xcc != (b % 8)
res = {"blocks": blocks, "spin_us": spin, "ms": ms, "xcds_seen": sorted(set(xcc.tolist())), "mismatch_fraction": float(mis.mean()),
       "mismatch_by_tenth_of_the_launch": [float(mis[i * blocks // 10:(i + 1) * blocks // 10].mean()) for i in range(10)],
       "first_mismatch_at_block": int(np.argmax(mis)) if mis.any() else None,
       "blocks_per_xcd": np.bincount(xcc, minlength=8).tolist()}
print(json.dumps(res))
