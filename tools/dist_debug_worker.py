"""Diagnostic twin of tests/dist_worker.py (native exchange over the stand-in library): on a wrong factor, every rank
reports the first supernodes whose columns differ from the oracle, with owner / group and where in the front."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from suitesparse_amd import cholmod as ch
    from suitesparse_amd import generators as G
    from oracle.oracle import OracleFactor
    n, Ap, Ai, Ax = G.poisson3d(32); perm = G.geometric_nd(32, 32, 32, 4)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    O.factorize(Ax)
    S = ch.Session(rank=rank, world=world, allreduce=None, factor_on_device=False, hip_flags=int(os.environ.get("CHOLMOD_TEST_HIP_FLAGS", "0")))
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.L.cholmod_l_hip_prepare(Lf, C.byref(S.cm)) == 1
    idb = np.zeros(128, dtype=np.uint8)
    if rank == 0:
        assert S.L.cholmod_hip_rccl_unique_id(idb.ctypes.data) == 0
    box = [idb.tobytes()]
    dist.broadcast_object_list(box, src=0)
    idb = np.frombuffer(box[0], dtype=np.uint8).copy()
    assert S.L.cholmod_hip_rccl_attach(ch.FactorView(Lf).hip_plan, idb.ctypes.data) == 0
    S.cm.error_handler = ch.ERRFUNC(0)
    ok = S.factorize(A, Lf)
    fv = ch.FactorView(Lf)
    res = {"rank": rank, "ok": int(ok), "status": int(S.cm.status), "bad": []}
    if fv.x is not None and len(fv.x):
        owner = np.empty(fv.nsuper, dtype=np.int64); S.L.cholmod_hip_get_partition(fv.hip_plan, owner.ctypes.data)
        g0 = np.empty(fv.nsuper, dtype=np.int64); gn = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_groups(fv.hip_plan, g0.ctypes.data, gn.ctypes.data)
        sp = np.empty(fv.nsuper, dtype=np.int64); lv = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_maps(fv.hip_plan, sp.ctypes.data, lv.ctypes.data, None)
        sup, pi, px = fv.super, fv.pi, fv.px
        for s in range(fv.nsuper):
            nscol, nsrow = int(sup[s + 1] - sup[s]), int(pi[s + 1] - pi[s])
            a = fv.x[px[s]:px[s] + nsrow * nscol].reshape(nscol, nsrow).T
            b = O.x[px[s]:px[s] + nsrow * nscol].reshape(nscol, nsrow).T
            d = np.abs(np.tril(a - b, 0) if True else a - b)
            d = np.abs(a - b)
            for j in range(nscol):
                d[:j, j] = 0
            if d.max() > 1e-9 * max(1.0, np.abs(b).max()):
                cols = np.where(d.max(axis=0) > 1e-9)[0]
                rows = np.where(d.max(axis=1) > 1e-9)[0]
                res["bad"].append(dict(s=s, level=int(lv[s]), owner=int(owner[s]), g0=int(g0[s]), gn=int(gn[s]), nscol=nscol, nsrow=nsrow,
                                       first_bad_col=int(cols[0]), last_bad_col=int(cols[-1]), nbadcols=int(len(cols)),
                                       first_bad_row=int(rows[0]), last_bad_row=int(rows[-1]), nbadrows=int(len(rows)), maxdiff=float(d.max()),
                                       nan=int(np.isnan(a).sum())))
                if len(res["bad"]) == 1:
                    rr, cc = np.where(d > 1e-9)
                    res["first_pattern"] = [(int(r_), int(c_), float(a[r_, c_]), float(b[r_, c_])) for r_, c_ in list(zip(rr, cc))[:60]]
                    res["first_children"] = [dict(s=int(c), nscol=int(sup[c + 1] - sup[c]), nsrow=int(pi[c + 1] - pi[c]), owner=int(owner[c])) for c in np.where(sp == s)[0]]
                if len(res["bad"]) >= 3:
                    break
    with open(f"{out}.{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


main()
