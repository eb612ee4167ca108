#!/usr/bin/env python3
"""Fingerprints of the engine's plans (cholmod_hip_debug_schedule_hash) over a battery of problems, worlds, ranks,
plan flags and tuning knobs -- host-only plans, no GPU.  `python tools/schedule_fingerprints.py write` records them in
tests/golden/schedule_fingerprints.json; tests/test_schedule_fingerprint.py compares.  Record them BEFORE restructuring
the scheduler / the plan builder, compare after: a refactoring must not move a single launch."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "schedule_fingerprints.json")

KNOBS = ["CHOLMOD_HIP_SHARE_AS_WORLD", "CHOLMOD_HIP_ARENA_BUDGET_MB", "CHOLMOD_HIP_OB1024_ROWS", "CHOLMOD_HIP_OB2048_ROWS",
         "CHOLMOD_HIP_OB4096_ROWS", "CHOLMOD_HIP_NO_CB_PASSTHROUGH", "CHOLMOD_HIP_NO_DISTRIBUTED_FRONTS", "CHOLMOD_HIP_OWN_W",
         "CHOLMOD_HIP_SHARED_CHAIN64", "CHOLMOD_HIP_CHAINF_AUTO", "CHOLMOD_HIP_UPD3_MIN_TILES",
         "CHOLMOD_HIP_NO_CHAINF", "CHOLMOD_HIP_NO_SUBGROUPS", "CHOLMOD_HIP_NO_CB_BALANCE",
         "CHOLMOD_HIP_UPD3_HALF_MAX"]


def problems():
    from suitesparse_amd import generators as G
    yield "p3d_24", G.poisson3d(24), G.geometric_nd(24, 24, 24, 4)
    yield "p3d_40", G.poisson3d(40), G.geometric_nd(40, 40, 40, 4)
    yield "box12r2", G.box_stencil3d(12, 2), G.geometric_nd(12, 12, 12, 3)
    yield "p2d_150", G.poisson2d(150), G.geometric_nd(150, 150, 1, 4)


def cases():
    # (world, ranks, flags, env)
    base = [(1, [0], 0, {})]
    for fl in (128, 512, 1024, 512 | 1024, 8192, 4, 2048):
        base.append((1, [0], fl, {}))
    base.append((1, [0], 0, {"CHOLMOD_HIP_UPD3_MIN_TILES": "16"}))
    base.append((1, [0], 0, {"CHOLMOD_HIP_UPD3_HALF_MAX": "0"}))                                            # whole tiles only
    base.append((1, [0], 0, {"CHOLMOD_HIP_OB1024_ROWS": "300", "CHOLMOD_HIP_OB2048_ROWS": "900", "CHOLMOD_HIP_OB4096_ROWS": "100000"}))
    base.append((1, [0], 0, {"CHOLMOD_HIP_ARENA_BUDGET_MB": "3"}))
    base.append((1, [0], 0, {"CHOLMOD_HIP_CHAINF_AUTO": "1"}))
    base.append((1, [0], 0, {"CHOLMOD_HIP_SHARE_AS_WORLD": "4"}))
    base.append((1, [0], 256, {"CHOLMOD_HIP_SHARE_AS_WORLD": "8", "CHOLMOD_HIP_OB1024_ROWS": "300"}))
    for world in (2, 3, 4, 8):
        ranks = list(range(world))
        base.append((world, ranks, 0, {}))
        base.append((world, ranks, 256, {}))
        base.append((world, ranks, 128, {"CHOLMOD_HIP_OWN_W": "64"}))
        base.append((world, ranks, 0, {"CHOLMOD_HIP_SHARED_CHAIN64": "1"}))
        base.append((world, ranks, 0, {"CHOLMOD_HIP_NO_CB_PASSTHROUGH": "1"}))
        base.append((world, ranks, 0, {"CHOLMOD_HIP_NO_DISTRIBUTED_FRONTS": "1"}))
        base.append((world, ranks, 0, {"CHOLMOD_HIP_OB1024_ROWS": "300", "CHOLMOD_HIP_OB2048_ROWS": "900"}))
        base.append((world, ranks, 0, {"CHOLMOD_HIP_ARENA_BUDGET_MB": "3"}))
        base.append((world, ranks, 0, {"CHOLMOD_HIP_NO_CB_BALANCE": "1", "CHOLMOD_HIP_NO_SUBGROUPS": "1"}))
    return base


def fingerprints():
    from suitesparse_amd import cholmod as ch
    out = {}
    saved = {k: os.environ.pop(k, None) for k in KNOBS}
    try:
        for pname, (n, Ap, Ai, Ax), perm in problems():
            S = ch.Session(use_gpu=0)
            A = S.sparse(n, Ap, Ai, Ax, -1)
            Lf = S.analyze(A, perm)
            fv = ch.FactorView(Lf)
            f = Lf.contents
            for world, ranks, flags, env in cases():
                for k, v in env.items():
                    os.environ[k] = v
                try:
                    for r in ranks:
                        st = C.c_int(0)
                        plan = S.L.cholmod_hip_plan_create_dist(fv.n, fv.nsuper, f.super, f.pi, f.px, f.s,
                                                                ch.HIP_PLAN_HOST_ONLY | flags, r, world, C.byref(st))
                        assert plan and st.value == 0, (pname, world, r, flags, env, st.value)
                        h = (C.c_uint64 * 16)()
                        assert S.L.cholmod_hip_debug_schedule_hash(plan, h) == 0
                        key = f"{pname}|w{world}|r{r}|f{flags}|" + ",".join(f"{k[12:]}={v}" for k, v in sorted(env.items()))
                        out[key] = [f"{x:016x}" for x in h]
                        S.L.cholmod_hip_plan_destroy(plan)
                finally:
                    for k in env:
                        os.environ.pop(k, None)
            S.free_factor(Lf); S.free_sparse(A); S.finish()
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
    return out


if __name__ == "__main__":
    fp = fingerprints()
    if len(sys.argv) > 1 and sys.argv[1] == "write":
        json.dump(fp, open(OUT, "w"), indent=0, sort_keys=True)
        print("wrote", len(fp), "fingerprints to", OUT)
    else:
        ref = json.load(open(OUT))
        bad = [k for k in ref if fp.get(k) != ref[k]]
        extra = [k for k in fp if k not in ref]
        print(len(ref), "reference fingerprints,", len(bad), "differ,", len(extra), "new")
        names = ["zg", "eg", "pg", "tg", "gg", "dg", "rg", "wg", "cg", "sm", "launches", "fr", "child", "relpairs", "scalars", "layout"]
        for k in bad[:20]:
            d = [names[i] for i in range(16) if k in fp and fp[k][i] != ref[k][i]]
            print("  ", k, "->", d)
        sys.exit(1 if bad else 0)
