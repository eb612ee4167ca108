#!/usr/bin/env python3
"""One-way hand-off time between two workgroups of one launch (probe k_handoff): payload sizes of a 64 x 64 block and a
few words, both blocks on one XCD (0, 8) or on two (0, 1), three coherence variants (agent-scope atomics as k_chainf;
workgroup-scope atomics = past the L1 into the XCD's L2; plain accesses behind a workgroup acquire)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
pr = ch.probes()
pr.cholmod_hip_bench_handoff.restype = C.c_double
pr.cholmod_hip_bench_handoff.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)]
out = []
for ndbl in (8, 4096, 16384):
    for (p, c, where) in ((0, 8, "same XCD"), (0, 1, "two XCDs"), (0, 16, "same XCD, 16 apart"), (0, 4, "two XCDs, 4 apart")):
        for mode, name in ((0, "agent"), (1, "workgroup-scope atomics"), (2, "plain + workgroup acquire")):
            bad = C.c_int(0)
            us = pr.cholmod_hip_bench_handoff(mode, p, c, ndbl, 300, C.byref(bad))
            out.append(dict(doubles=ndbl, blocks=[p, c], where=where, mode=name, one_way_us=round(us, 3), stale_or_timeout=bad.value))
            print(out[-1])
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06_handoff.json"), "w"), indent=1)
