"""Worker for the multi-rank tests: one process per rank (gloo rendezvous on
127.0.0.1).  mode "gpu": every rank drives the engine (ranks may share GPU 0),
factorizes a partitioned problem and checks its gathered factor against the
oracle.  mode "cpu": exercises the all-reduce callback on host memory."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, case, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    backend = os.environ.get("DIST_TEST_BACKEND", "gloo")
    if backend == "nccl":       # RCCL: one rank per GPU (the 1-GPU box allows world 1 only)
        import torch
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from suitesparse_amd import cholmod as ch
    from suitesparse_amd import generators as G
    from suitesparse_amd.dist import make_allreduce
    res = {"rank": rank}
    if mode == "cpu":
        cb = make_allreduce(device_memory=False)
        buf = np.arange(10, dtype=np.float64) * (rank + 1)
        assert cb(buf.ctypes.data, buf.size, 0, world, None) == 0
        res["ok"] = bool(np.allclose(buf, np.arange(10) * sum(range(1, world + 1))))
        res["calls"] = cb.stats["n"]
    else:
        from oracle.oracle import OracleFactor, bind_blas
        if case.startswith("checks_p3d_"):
            # a BASELINE-size problem no oracle can follow in a test (Poisson 100^3: configs[1]) on `world` real peers: the
            # factor stays distributed; every rank checks its own part (cholmod_hip_factor_checks_local), the five sums meet
            # in an all-reduce and are held against the closed forms -- log det A of the Dirichlet Laplacian and
            # ||L||_F^2 = trace A -- after the first factorization and after a resident refactorization
            import ctypes as C
            import torch
            m = int(case.split("_")[-1])
            n, Ap, Ai, Ax = G.poisson3d(m); perm = G.geometric_nd(m, m, m, 4)
            S = ch.Session(rank=rank, world=world, allreduce=None, factor_on_device=True)
            A = S.sparse(n, Ap, Ai, Ax, -1)
            Lf = S.analyze(A, perm)
            assert S.L.cholmod_l_hip_prepare(Lf, C.byref(S.cm)) == 1, S.cm.status
            idb = np.zeros(128, dtype=np.uint8)
            if rank == 0:
                assert S.L.cholmod_hip_rccl_unique_id(idb.ctypes.data) == 0
            box = [idb.tobytes()]
            dist.broadcast_object_list(box, src=0)
            idb = np.frombuffer(box[0], dtype=np.uint8).copy()
            assert S.L.cholmod_hip_rccl_attach(ch.FactorView(Lf).hip_plan, idb.ctypes.data) == 0
            ld = G.poisson_logdet(m, m, m)
            errs = []
            for it in range(2):
                ok = S.factorize(A, Lf) if it == 0 else S.refactorize_resident(Lf)
                assert ok == 1 and S.cm.status == ch.OK, (ok, S.cm.status)
                loc = torch.from_numpy(np.asarray(S.factor_checks_local(Lf), dtype=np.float64).copy())
                dist.all_reduce(loc)
                v = loc.numpy()
                errs.append(dict(logdet_rel_err=float(abs(2.0 * v[0] - ld) / abs(ld)), upper_nonzeros=int(v[1]), nonfinite=int(v[2]),
                                 trace_rel_err=float(abs(v[3] - 6.0 * n) / (6.0 * n)), nonpositive_diag=int(v[4])))
            st = S.hip_stats(Lf)
            res.update(ok=1, status=int(S.cm.status), checks=errs, exchanges=int(st[17]), L_bytes_rank=float(st[36]),
                       L_bytes_whole=float(st[5]), arena_bytes=float(st[4]))
            S.free_factor(Lf)
            S.free_sparse(A)
            S.finish()
            with open(f"{out}.{rank}", "w") as f:
                json.dump(res, f)
            dist.barrier()
            dist.destroy_process_group()
            return
        if case in ("p3d_64", "p3d_32_notposdef_root") or (case in ("p3d_48", "p3d_40") and world >= 8):
            bind_blas()                 # (the oracle's dense kernels through a BLAS: 64^3 in seconds instead of minutes)
        if case == "p3d_20":
            n, Ap, Ai, Ax = G.poisson3d(20); perm = G.geometric_nd(20, 20, 20, 4)
        elif case == "p3d_32":
            n, Ap, Ai, Ax = G.poisson3d(32); perm = G.geometric_nd(32, 32, 32, 4)
        elif case == "p3d_40":
            n, Ap, Ai, Ax = G.poisson3d(40); perm = G.geometric_nd(40, 40, 40, 4)
        elif case == "p3d_48":
            n, Ap, Ai, Ax = G.poisson3d(48); perm = G.geometric_nd(48, 48, 48, 4)
        elif case == "p3d_64":
            n, Ap, Ai, Ax = G.poisson3d(64); perm = G.geometric_nd(64, 64, 64, 4)
        elif case == "p3d_32_notposdef_root":
            n, Ap, Ai, Ax = G.poisson3d(32); perm = G.geometric_nd(32, 32, 32, 4)
        elif case == "p2d_90":
            n, Ap, Ai, Ax = G.poisson2d(90); perm = G.geometric_nd(90, 90, 1, 4)
        elif case == "box10":
            n, Ap, Ai, Ax = G.box_stencil3d(10, 2); perm = G.geometric_nd(10, 10, 10, 3)
        elif case == "dense_1400":
            # one dense supernode: the root IS the factor, shared by everybody -- its window's virtual base
            # (window start minus outer-block offset) lies below the rank's array from the second outer block on
            rng = np.random.default_rng(7)
            n = 1400
            M = np.tril(rng.standard_normal((n, n)))
            M[np.arange(n), np.arange(n)] = 2.0 * n ** 0.5 + rng.random(n)
            import scipy.sparse as sp
            Asp = sp.csc_matrix(M)
            Asp.sort_indices()
            Ap, Ai, Ax = Asp.indptr.astype(np.int64), Asp.indices.astype(np.int64), Asp.data.astype(np.float64)
            perm = np.arange(n, dtype=np.int64)
        elif case in ("p3d_16_notposdef", "p3d_16_notposdef_root"):
            n, Ap, Ai, Ax = G.poisson3d(16); perm = G.geometric_nd(16, 16, 16, 4)
        elif case == "p3d_20_complex":
            pass
        else:
            raise KeyError(case)
        if case == "p3d_20_complex":
            # complex (Hermitian) input on several ranks: the real twin of the factor (every row / column doubled) with the
            # even-column update kernels, its shared fronts distributed by slabs like any other (host/complex.c)
            from tests.test_complex import hermitian_from, full_hermitian
            n, Ap, Ai, Ax = G.poisson3d(20); perm = G.geometric_nd(20, 20, 20, 4)
            Ap, Ai, Ax = hermitian_from(n, Ap, Ai, Ax, seed=3)
            O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
            assert O.factorize_complex(Ax) == 0
            S = ch.Session(rank=rank, world=world, allreduce=make_allreduce(), hip_flags=int(os.environ.get("CHOLMOD_TEST_HIP_FLAGS", "0")))
            A = S.sparse(n, Ap, Ai, Ax, -1)
            Lf = S.analyze(A, perm)
            ok = S.factorize(A, Lf)
            fv = ch.FactorView(Lf)
            m = O.lower_mask()
            rng = np.random.default_rng(5)
            b = rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))
            x = S.solve(Lf, b)
            res.update(ok=int(ok), status=int(S.cm.status), err=float(np.linalg.norm((fv.x - O.xc)[m]) / np.linalg.norm(O.xc[m])),
                       upper_zero=bool(np.all(fv.x[~m] == 0)),
                       resid=float(np.linalg.norm(full_hermitian(n, Ap, Ai, Ax) @ x.T - b.T) / np.linalg.norm(b)))
            S.free_factor(Lf)
            S.free_sparse(A)
            S.finish()
            with open(f"{out}.{rank}", "w") as f:
                json.dump(res, f)
            dist.barrier()
            dist.destroy_process_group()
            return
        import time
        t_w0 = time.perf_counter()
        O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
        if "notposdef" in case:
            sup = O.super
            cand = [s for s in range(O.nsuper // 3, O.nsuper) if sup[s + 1] - sup[s] >= 6]
            kbad = int(sup[cand[0]] + 2)
            if case.endswith("_root"):
                # a column in the middle of the last supernode (the root: shared by every rank), past its first 64-column step
                kbad = int(sup[O.nsuper - 1] + min(100, (sup[O.nsuper] - sup[O.nsuper - 1]) // 2))
            Ax = Ax.copy()
            Ax[Ap[int(O.Perm[kbad])]] = -3.0
        st_o = O.factorize(Ax)
        t_w1 = time.perf_counter()
        # exchange: the torch.distributed callback (gloo staging), or -- DIST_TEST_EXCHANGE=native --
        # the engine's own path (cholmod_hip_rccl_attach: communicators, splits, stream-ordered
        # reduce-scatter / all-gather / all-reduce) on the collective library named by
        # CHOLMOD_HIP_RCCL_LIBRARY (tests/standin_rccl: ranks sharing one GPU)
        native = os.environ.get("DIST_TEST_EXCHANGE") == "native"
        cb = None if native else make_allreduce()
        resident = os.environ.get("DIST_TEST_RESIDENT") == "1"
        S = ch.Session(rank=rank, world=world, allreduce=cb, factor_on_device=resident or os.environ.get("DIST_TEST_RESIDENT_PLAIN") == "1",
                       hip_flags=int(os.environ.get("CHOLMOD_TEST_HIP_FLAGS", "0")))
        A = S.sparse(n, Ap, Ai, Ax, -1)
        Lf = S.analyze(A, perm)
        if native:
            import ctypes as C
            assert S.L.cholmod_l_hip_prepare(Lf, C.byref(S.cm)) == 1, S.cm.status
            idb = np.zeros(128, dtype=np.uint8)
            if rank == 0:
                assert S.L.cholmod_hip_rccl_unique_id(idb.ctypes.data) == 0
            box = [idb.tobytes()]
            dist.broadcast_object_list(box, src=0)
            idb = np.frombuffer(box[0], dtype=np.uint8).copy()
            assert S.L.cholmod_hip_rccl_attach(ch.FactorView(Lf).hip_plan, idb.ctypes.data) == 0
        t_w2 = time.perf_counter()
        ok = S.factorize(A, Lf)
        t_w3 = time.perf_counter()
        res["seconds"] = {"oracle": t_w1 - t_w0, "analyze_attach": t_w2 - t_w1, "factorize_gather_download": t_w3 - t_w2}
        if resident:
            # factor left distributed on the devices: a second factorization clears only the
            # slabs of Lx this rank holds; the gather (factor_to_host) then fills the rest, and
            # the factorization after it must clear everything again
            import ctypes as C
            m0 = O.lower_mask()
            errs = []
            for _ in range(2):
                assert S.refactorize_resident(Lf) == 1
                assert S.L.cholmod_l_gather_factor(Lf, C.byref(S.cm)) == 1
                assert S.L.cholmod_l_factor_to_host(Lf, C.byref(S.cm)) == 1
                x = ch.FactorView(Lf).x
                errs.append(float(np.linalg.norm((x - O.x)[m0]) / np.linalg.norm(O.x[m0])))
            res["resident_errs"] = errs
            assert S.refactorize_resident(Lf) == 1
            assert S.L.cholmod_l_gather_factor(Lf, C.byref(S.cm)) == 1
            assert S.L.cholmod_l_factor_to_host(Lf, C.byref(S.cm)) == 1
        if os.environ.get("CHOLMOD_HIP_TEST_GATHER_STAGED"):
            # one rank has no room for the complete factor NEXT TO its own part (two ranks at the headline size): its
            # part takes the detour through host memory; the gathered factor must be right, and so must the next
            # factorization (which reserves the rank's own array again) and its gather
            import ctypes as C
            assert ok == 1
            m0 = O.lower_mask()
            errs = []
            for it in range(2):
                assert S.L.cholmod_l_gather_factor(Lf, C.byref(S.cm)) == 1, S.cm.status
                assert S.L.cholmod_l_factor_to_host(Lf, C.byref(S.cm)) == 1
                x = ch.FactorView(Lf).x
                errs.append(float(np.linalg.norm((x - O.x)[m0]) / np.linalg.norm(O.x[m0])))
                b = G.demo_rhs(n)
                xs = S.solve(Lf, b)
                errs.append(float(np.linalg.norm(G.sym_matvec(n, Ap, Ai, Ax, -1, xs) - b) / np.linalg.norm(b)))
                if it == 0:
                    assert S.refactorize_resident(Lf) == 1
            res.update(staged_errs=errs)
            S.free_factor(Lf)
            S.free_sparse(A)
            S.finish()
            with open(f"{out}.{rank}", "w") as f:
                json.dump(res, f)
            dist.barrier()
            dist.destroy_process_group()
            return
        if os.environ.get("CHOLMOD_HIP_TEST_FAIL_GATHER"):
            # one rank has no room for the complete factor: EVERY rank's gather must come back with
            # CHOLMOD_OUT_OF_MEMORY (nobody left waiting in the collective); the distributed factor survives:
            # without the hook the next gather succeeds and the factor matches the oracle
            import ctypes as C
            assert ok == 1
            S.cm.error_handler = ch.ERRFUNC(0)
            g1 = S.L.cholmod_l_gather_factor(Lf, C.byref(S.cm))
            st1 = int(S.cm.status)
            del os.environ["CHOLMOD_HIP_TEST_FAIL_GATHER"]
            S.cm.status = ch.OK
            g2 = S.L.cholmod_l_gather_factor(Lf, C.byref(S.cm))
            assert S.L.cholmod_l_factor_to_host(Lf, C.byref(S.cm)) == 1
            m0 = O.lower_mask()
            x = ch.FactorView(Lf).x
            res.update(gather_failed=int(g1), gather_failed_status=st1, gather_again=int(g2),
                       err=float(np.linalg.norm((x - O.x)[m0]) / np.linalg.norm(O.x[m0])))
            S.free_factor(Lf)
            S.free_sparse(A)
            S.finish()
            with open(f"{out}.{rank}", "w") as f:
                json.dump(res, f)
            dist.barrier()
            dist.destroy_process_group()
            return
        if os.environ.get("CHOLMOD_HIP_TEST_FAIL_LAUNCH"):
            # failure-injection case: every rank must come back with an error (no hang)
            res.update(ok=int(ok), status=int(S.cm.status))
            S.free_factor(Lf)
            S.free_sparse(A)
            S.finish()
            with open(f"{out}.{rank}", "w") as f:
                json.dump(res, f)
            dist.barrier()
            dist.destroy_process_group()
            return
        fv = ch.FactorView(Lf)
        owner = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_partition(fv.hip_plan, owner.ctypes.data)
        m = O.lower_mask()
        err = float(np.linalg.norm((fv.x - O.x)[m]) / np.linalg.norm(O.x[m]))
        res.update(ok=int(ok), status=int(S.cm.status), oracle_status=int(st_o), err=err,
                   minor=int(fv.minor), oracle_minor=int(O.minor),
                   zero_pattern_equal=bool(np.array_equal(fv.x[m] != 0, O.x[m] != 0)),
                   nshared=int((owner < 0).sum()), nsuper=int(fv.nsuper),
                   owned=[int((owner == r).sum()) for r in range(world)],
                   allreduce_calls=cb.stats["n"] if cb else int(S.hip_stats(Lf)[17]),
                   allreduce_MB=(cb.stats["bytes"] if cb else S.hip_stats(Lf)[18]) / 1e6,
                   allreduce_group_sizes=sorted(cb.stats["by_size"]) if cb else [],
                   nsplit=int(S.hip_stats(Lf)[22]), window_opens=int(S.hip_stats(Lf)[37]),
                   window_opens_negative_base=int(S.hip_stats(Lf)[38]), L_bytes_rank=float(S.hip_stats(Lf)[36]),
                   L_bytes_whole=float(S.hip_stats(Lf)[5]),
                   gather_MB=float(S.hip_stats(Lf)[25]) / 1e6, gather_inline_MB=float(S.hip_stats(Lf)[39]) / 1e6)
        if st_o == 0:
            # the rank's share of the factor invariants (its own part of L, no gathered copy) and the invariants
            # of the gathered factor: the shares must add up to them
            res["checks_local"] = S.factor_checks_local(Lf).tolist()
            fc = S.factor_checks(Lf)
            res["checks_full"] = [fc["half_logdet"], fc["upper_nonzeros"], fc["nonfinite"], fc["fro2"], fc["nonpositive_diag"]]
        if native:
            g0 = np.empty(fv.nsuper, dtype=np.int64)
            gn = np.empty(fv.nsuper, dtype=np.int64)
            assert S.L.cholmod_hip_get_groups(fv.hip_plan, g0.ctypes.data, gn.ctypes.data) == 0
            res["allreduce_group_sizes"] = sorted({int(x) for x in gn[gn > 1]})
        if st_o == 0:
            b = G.demo_rhs(n)
            x = S.solve(Lf, b)
            r = G.sym_matvec(n, Ap, Ai, Ax, -1, x) - b
            res["resid"] = float(np.linalg.norm(r) / np.linalg.norm(b))
        S.free_factor(Lf)
        S.free_sparse(A)
        S.finish()
    with open(f"{out}.{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
