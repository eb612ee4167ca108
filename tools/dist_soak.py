"""Soak of the distributed path on a one-GPU box: 36 factorizations with 2 - 4 ranks sharing GPU 0 over the asynchronous
stand-in collective library, arena and windows poisoned, every rank against the oracle.  usage: python tools/dist_soak.py [jitter]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import tests.test_dist as T
env = dict(T.NATIVE, CHOLMOD_HIP_TEST_POISON_ARENA="1", SSAMD_TEST_HOOKS_LIB="1")
bad = 0
JIT = len(sys.argv) > 1 and sys.argv[1] == "jitter"        # random hold-ups of the streams (CHOLMOD_HIP_TEST_JITTER), a new seed per round
for it in range(12):
    for world, case in ((4, "p3d_32"), (3, "p3d_48"), (2, "dense_1400")):
        ex = dict(env, CHOLMOD_HIP_UPD3_MIN_TILES="1" if it % 2 else "2048")
        # every third round: outer blocks of four block columns, so that far-row gathers ride the exchange stream beside the chain
        if it % 3 == 1: ex.update(T.WIDE_OUTER)
        if JIT: ex["CHOLMOD_HIP_TEST_JITTER"] = f"{100 + it}:{500 if it % 3 else 3000}"
        res = T._run_ranks(world, "gpu", case, extra_env=ex)
        for r in res:
            if not (r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11):
                bad += 1; print("BAD", it, world, case, r, flush=True)
print("soak: 36 distributed factorizations x ranks, failures:", bad, flush=True)
