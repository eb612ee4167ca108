"""What one rank of a W-GPU run computes, measured on a 1-GPU box.

The engine is driven as rank r of world W with an all-reduce callback that
moves no data (the numbers in the shared fronts are then meaningless, the
launch list and its timing are exactly rank r's).  Prints per-class device
seconds so the replicated (Amdahl) part of the multi-GPU schedule is a
measurement, not a model:  T_W(r) = compute of rank r, excluding the xGMI time
of the all-reduces themselves (volume is printed to price it).

    python tools/dist_whatif.py --grid 100 --world 8 --ranks 0,3
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=100)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", default="0")
    ap.add_argument("--hip-flags", type=int, default=0)
    args = ap.parse_args()
    import torch
    from suitesparse_amd import cholmod as ch
    from suitesparse_amd import generators as G
    from suitesparse_amd.dist import _DevView

    m, W = args.grid, args.world
    n, Ap, Ai, Ax = G.poisson3d(m)
    perm = G.geometric_nd(m, m, m, 4)
    for r in [int(x) for x in args.ranks.split(",")]:
        calls = {"n": 0, "bytes": 0, "by_size": {}}

        def _fn(ptr, count, first, size, user):
            calls["n"] += 1
            calls["bytes"] += 8 * int(count)
            calls["by_size"][int(size)] = calls["by_size"].get(int(size), 0) + 8e-9 * int(count)
            if count in (2 * W, 3 * W):          # the not-posdef / failure agreement: "nobody failed"
                t = torch.as_tensor(_DevView(ptr, count), device="cuda")
                t[:W] = 1e18
                torch.cuda.synchronize()
            return 0

        cb = ch.ALLREDUCE_FN(_fn)
        S = ch.Session(factor_on_device=True, hip_flags=args.hip_flags, rank=r, world=W, allreduce=cb)
        A = S.sparse(n, Ap, Ai, Ax, -1)
        Lf = S.analyze(A, perm)
        fl = S.cm.fl
        S.factorize(A, Lf)
        calls["n"] = calls["bytes"] = 0
        calls["by_size"] = {}
        t0 = time.perf_counter()
        S.refactorize_resident(Lf)
        wall = time.perf_counter() - t0
        ncall, nbytes, by_size = calls["n"], calls["bytes"], dict(calls["by_size"])
        S.set_profiling(Lf, True)
        S.refactorize_resident(Lf)
        ps = S.hip_stats(Lf)
        S.set_profiling(Lf, False)
        if os.environ.get("WHATIF_TOP"):
            # the longest launches of the kinds named (e.g. WHATIF_TOP=1,15: extend-add and window moves)
            lp = S.launch_profile(Lf)
            kinds = [int(k) for k in os.environ["WHATIF_TOP"].split(",")]
            idx = [i for i in np.argsort(-lp["ms"]) if lp["kind"][i] in kinds][:25]
            for i in idx:
                print("   launch %5d kind %2d grid %7d ms %8.3f MB %9.1f" % (i, lp["kind"][i], lp["grid"][i], lp["ms"][i], lp["bytes"][i] / 1e6), file=sys.stderr)
            for k in kinds:
                q = lp["kind"] == k
                print("   kind %d: %d launches %.3f s" % (k, int(q.sum()), 1e-3 * float(lp["ms"][q].sum())), file=sys.stderr)
        fv = ch.FactorView(Lf)
        owner = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_partition(fv.hip_plan, owner.ctypes.data)
        print(json.dumps({
            "grid": m, "world": W, "rank": r, "wall_ms_no_comm": 1e3 * wall, "fl": fl,
            "exec_flops_this_rank": ps[1], "launches": int(ps[2]),
            "shared_fronts": int((owner < 0).sum()), "own_fronts": int((owner == r).sum()),
            "allreduce_calls": ncall, "allreduce_GB": 1e-9 * nbytes, "allreduce_GB_by_group_size": by_size,
            "gather_GB": 1e-9 * ps[25], "gather_GB_waited_for_where_issued": 1e-9 * ps[39],
            "L_GB_this_rank": 1e-9 * ps[36], "L_GB_whole": 1e-9 * ps[5], "arena_GB_this_rank": 1e-9 * ps[4],
            "profiled_seconds": {"update_wave_tiles": ps[32], "update_wave_tiles_TF": 1e-12 * ps[34] / max(ps[32], 1e-30),
                                 "update64": ps[6], "update64_TF": 1e-12 * ps[8] / max(ps[6], 1e-30),
                                 "update64_plus_potrf": ps[27], "trsm_upd": ps[30],
                                 "extend_add+zero": ps[9], "potrf": ps[11], "trsm": ps[12],
                                 "assemble": ps[13], "small_fronts": ps[19], "exchanges": ps[17],
                                 "total": ps[0]}}), flush=True)
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()


if __name__ == "__main__":
    main()
