"""Multi-rank path: ownership partition, the all-reduce callback
over gloo (CPU, world_size 2/3), and -- on the GPU box -- full distributed
factorizations with several ranks sharing GPU 0 through the gloo-staged
exchange, each rank checking its gathered factor against the oracle."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from suitesparse_amd import cholmod as ch
from suitesparse_amd import generators as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))     # (tools/dist_soak.py imports this module as tests.test_dist)
from ports import free_port as _free_port  # noqa: E402


def _run_ranks(world, mode, case, timeout=600, extra_env=None):
    port = _free_port()
    out = os.path.join(tempfile.mkdtemp(), "res")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
        # (several processes on one host: the cores are shared out -- eight ranks that each start a full-width OpenMP /
        # OpenBLAS team for the oracle and the analysis spend their time spinning on each other)
        share = str(max(1, (os.cpu_count() or 1) // world))
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            env.setdefault(k, share)
        env.update(extra_env or {})
        if any(k.startswith("CHOLMOD_HIP_TEST_") for k in env):
            env["SSAMD_TEST_HOOKS_LIB"] = "1"      # the engine's test hooks exist only in lib/libcholmod_amd_testhooks.so
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"),
                                       mode, case, out], env=env, cwd=ROOT))
    rcs = [p.wait(timeout=timeout) for p in procs]
    assert rcs == [0] * world, rcs
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


def _host_plans(world, n, Ap, Ai, Ax, perm, postorder=True, ranks=None, analysed=None):
    """Host-only plans of the given ranks of a world.  analysed = (S, A, Lf) of an earlier call (kept by the caller, who
    frees it): the analysis is not repeated."""
    if analysed is None:
        S = ch.Session(use_gpu=0, postorder=postorder)
        A = S.sparse(n, Ap, Ai, Ax, -1)
        Lf = S.analyze(A, perm)
    else:
        S, A, Lf = analysed
    fv = ch.FactorView(Lf)
    f = Lf.contents
    owners, stats, levels = [], [], None
    groups = _host_plans.groups = []
    for r in (range(world) if ranks is None else ranks):
        st = C.c_int(0)
        plan = S.L.cholmod_hip_plan_create_dist(fv.n, fv.nsuper, f.super, f.pi, f.px, f.s,
                                                ch.HIP_PLAN_HOST_ONLY, r, world, C.byref(st))
        assert plan and st.value == 0
        o = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_partition(plan, o.ctypes.data)
        g0 = np.empty(fv.nsuper, dtype=np.int64)
        gn = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_groups(plan, g0.ctypes.data, gn.ctypes.data)
        groups.append((g0, gn))
        sp = np.empty(fv.nsuper, dtype=np.int64)
        lv = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_maps(plan, sp.ctypes.data, lv.ctypes.data, None)
        s16 = np.zeros(ch.CHOLMOD_HIP_NSTATS)
        S.L.cholmod_hip_get_stats(plan, s16.ctypes.data)
        owners.append(o); stats.append(s16); levels = (sp, lv)
        S.L.cholmod_hip_plan_destroy(plan)
    nscol = np.diff(fv.super).astype(float)
    nsrow = np.diff(fv.pi).astype(float)
    if analysed is None:
        S.free_factor(Lf); S.free_sparse(A); S.finish()
    return owners, stats, levels, nscol, nsrow


@pytest.mark.parametrize("world", [2, 3, 8])
def test_partition_is_consistent_and_balanced(world):
    n, Ap, Ai, Ax = G.poisson3d(24)
    owners, stats, (sparent, level), nscol, nsrow = _host_plans(world, n, Ap, Ai, Ax,
                                                               G.geometric_nd(24, 24, 24, 4))
    o = owners[0]
    for other in owners[1:]:
        assert np.array_equal(o, other)              # every rank derives the same map
    assert o.min() == -1 and o.max() == world - 1
    shared = o < 0
    has_parent = sparent >= 0
    # shared fronts are closed towards the root; a solo front's parent is solo
    # on the same rank or shared
    assert np.all(shared[sparent[shared & has_parent]])
    solo = ~shared & has_parent
    par = sparent[solo]
    assert np.all((o[par] == o[solo]) | (o[par] < 0))
    # rank groups: identical on every rank, the roots' group is everybody, a
    # child's group lies inside its parent's, solo fronts are groups of one
    g0, gn = _host_plans.groups[0]
    for a, b in _host_plans.groups[1:]:
        assert np.array_equal(a, g0) and np.array_equal(b, gn)
    assert np.all(gn[shared] >= 2) and np.all(gn[~shared] == 1) and np.all(g0[~shared] == o[~shared])
    roots = shared & ~has_parent
    assert np.all((g0[roots] == 0) & (gn[roots] == world))
    hp = np.where(has_parent)[0]
    pa = sparent[hp]
    assert np.all((g0[hp] >= g0[pa]) & (g0[hp] + gn[hp] <= g0[pa] + gn[pa]))
    # load balance (flop weights as the engine uses): private subtrees plus an
    # equal share of the shared fronts' own work per group member
    ncb = nsrow - nscol
    w = nscol ** 3 / 3 + ncb * nscol ** 2 + ncb ** 2 * nscol
    loads = np.array([w[o == r].sum() + (w[shared & (g0 <= r) & (r < g0 + gn)]
                                         / gn[shared & (g0 <= r) & (r < g0 + gn)]).sum()
                      for r in range(world)])
    assert loads.max() <= 1.3 * loads.mean(), loads / loads.mean()
    # the executed-flop statistic is global, identical on every rank
    assert all(abs(s[1] - w.sum()) < 1e-9 * w.sum() for s in stats)


@pytest.mark.parametrize("world", [4, 8])
def test_batch_split_is_identical_on_every_rank(world):
    """The memory-aware batch split (engine.hip: build_host) is chosen from an arena laid out over ALL fronts, and the
    collectives of a group are issued in batch order: every rank must derive the same arena length, the same number of
    subtrees and the same batch for every front -- also where a shared front's group is a strict subset of the world (a
    member and a non-member once counted its distributed contribution block differently; round-4 advisor item).  Budgets
    swept across the whole range in which the decision changes."""
    n, Ap, Ai, Ax = G.poisson3d(24)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, G.geometric_nd(24, 24, 24, 4))
    fv = ch.FactorView(Lf)
    f = Lf.contents

    def plans(budget_mb):
        out = []
        old = os.environ.get("CHOLMOD_HIP_ARENA_BUDGET_MB")
        if budget_mb is not None:
            os.environ["CHOLMOD_HIP_ARENA_BUDGET_MB"] = repr(budget_mb)
        try:
            for r in range(world):
                st = C.c_int(0)
                plan = S.L.cholmod_hip_plan_create_dist(fv.n, fv.nsuper, f.super, f.pi, f.px, f.s,
                                                        ch.HIP_PLAN_HOST_ONLY, r, world, C.byref(st))
                assert plan and st.value == 0
                b = np.empty(fv.nsuper, dtype=np.int64)
                ga = C.c_int64(0)
                ns = S.L.cholmod_hip_get_batches(plan, b.ctypes.data, C.byref(ga))
                g0 = np.empty(fv.nsuper, dtype=np.int64); gn = np.empty(fv.nsuper, dtype=np.int64)
                S.L.cholmod_hip_get_groups(plan, g0.ctypes.data, gn.ctypes.data)
                out.append((ns, ga.value, b, gn))
                S.L.cholmod_hip_plan_destroy(plan)
        finally:
            if budget_mb is not None:
                if old is None:
                    del os.environ["CHOLMOD_HIP_ARENA_BUDGET_MB"]
                else:
                    os.environ["CHOLMOD_HIP_ARENA_BUDGET_MB"] = old
        return out

    base = plans(None)
    gn = base[0][3]
    assert ((gn > 1) & (gn < world)).any()          # the case in question: a group smaller than the world
    full_mb = 8.0 * base[0][1] / 1048576.0
    seen = set()
    for frac in [None] + [0.02 * k for k in range(1, 60)]:
        res = plans(None if frac is None else full_mb * frac)
        ns0, ga0, b0, _ = res[0]
        assert b0.min() >= 0
        for ns, ga, b, _ in res[1:]:
            assert ns == ns0 and ga == ga0 and np.array_equal(b, b0), (frac, ns, ns0, ga, ga0)
        seen.add(ns0)
    assert len(seen) >= 2, seen                     # the sweep did cross at least one split decision
    S.free_factor(Lf); S.free_sparse(A); S.finish()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_contribution_routing_is_complete_and_disjoint(world):
    """Contributions routed past the contribution blocks of shared fronts (host side, no GPU): on every rank, a front it holds
    whose parent is shared contributes to EVERY shared ancestor (one pair each, no pair for anybody else); over the ranks,
    the blocks of contribution-block columns of a shared front partition [0, ncb) among the members of its group."""
    n, Ap, Ai, Ax = G.poisson3d(24)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, G.geometric_nd(24, 24, 24, 4))
    fv = ch.FactorView(Lf)
    f = Lf.contents
    blocks, npairs_seen = {}, 0
    for r in range(world):
        st = C.c_int(0)
        plan = S.L.cholmod_hip_plan_create_dist(fv.n, fv.nsuper, f.super, f.pi, f.px, f.s, ch.HIP_PLAN_HOST_ONLY, r, world, C.byref(st))
        assert plan and st.value == 0
        o = np.empty(fv.nsuper, dtype=np.int64); S.L.cholmod_hip_get_partition(plan, o.ctypes.data)
        g0 = np.empty(fv.nsuper, dtype=np.int64); gn = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_groups(plan, g0.ctypes.data, gn.ctypes.data)
        sp = np.empty(fv.nsuper, dtype=np.int64); lv = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_get_maps(plan, sp.ctypes.data, lv.ctypes.data, None)
        npair = S.L.cholmod_hip_debug_routing(plan, 0, None, None, None, None)
        pd = np.empty(max(npair, 1), dtype=np.int64); pa = np.empty(max(npair, 1), dtype=np.int64)
        lo = np.empty(fv.nsuper, dtype=np.int64); hi = np.empty(fv.nsuper, dtype=np.int64)
        S.L.cholmod_hip_debug_routing(plan, npair, pd.ctypes.data, pa.ctypes.data, lo.ctypes.data, hi.ctypes.data)
        S.L.cholmod_hip_plan_destroy(plan)
        mine = (g0 <= r) & (r < g0 + gn)
        ncb = (fv.pi[1:] - fv.pi[:-1]) - (fv.super[1:] - fv.super[:-1])
        want = set()
        for d in np.where(mine & (sp >= 0))[0]:
            if o[sp[d]] >= 0 or ncb[d] == 0:
                continue                                        # parent private: the ordinary child -> parent extend-add
            a = sp[d]
            while a >= 0:
                assert o[a] < 0 and mine[a]                     # shared towards the root, and a child's group lies in its parent's
                want.add((int(d), int(a)))
                a = sp[a]
        got = list(zip(pd[:npair].tolist(), pa[:npair].tolist()))
        assert len(got) == len(set(got)) and set(got) == want, (r, len(got), len(want))
        npairs_seen += len(got)
        for s_ in np.where((o < 0) & mine & (ncb > 0))[0]:
            blocks.setdefault(int(s_), []).append((int(lo[s_]), int(hi[s_]), int(ncb[s_]), int(gn[s_])))
        assert np.all(lo[~((o < 0) & mine & (ncb > 0))] == -1)
    assert blocks and npairs_seen >= 2 * world
    for s_, bl in blocks.items():
        assert len(bl) == bl[0][3]                               # one block per member of the group
        bl.sort()
        assert bl[0][0] == 0 and bl[-1][1] == bl[0][2]
        assert all(bl[q][1] == bl[q + 1][0] for q in range(len(bl) - 1)) and all(b[0] <= b[1] and b[0] % 64 == 0 for b in bl)
    S.free_factor(Lf); S.free_sparse(A); S.finish()


def test_partition_balance_at_the_headline_size():
    """The metric's multi-GPU workload itself (Poisson 200^3, 8 M dof, geometric ND) through the
    host-only plan of every rank: the same partition everywhere and the flop loads the
    proportional mapping estimates within 5 % of the mean at 2, 4 and 8 ranks (measured:
    within 1.4 % at 8, 1.2 % at 4, 0.6 % at 2 ranks; a front is shared while its subtree
    outweighs 1 / (6 world) of the factorization)."""
    m = 200
    n, Ap, Ai, Ax = G.poisson3d(m)
    perm = G.geometric_nd(m, m, m, 4)
    S = ch.Session(use_gpu=0)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)                          # (once for the three worlds)
    for world in (8, 4, 2):
        owners, stats, (sparent, level), nscol, nsrow = _host_plans(world, n, Ap, Ai, Ax, perm,
                                                                   ranks=(0, world - 1), analysed=(S, A, Lf))
        o = owners[0]
        assert np.array_equal(o, owners[1])
        g0, gn = _host_plans.groups[0]
        shared = o < 0
        ncb = nsrow - nscol
        w = nscol ** 3 / 3 + ncb * nscol ** 2 + ncb ** 2 * nscol
        loads = np.array([w[o == r].sum() + (w[shared & (g0 <= r) & (r < g0 + gn)]
                                             / gn[shared & (g0 <= r) & (r < g0 + gn)]).sum()
                          for r in range(world)])
        print("world", world, "loads / mean", np.round(loads / loads.mean(), 4), "shared fronts", int(shared.sum()))
        for st in stats:
            print("   exchange GB", round(st[18] / 1e9, 2), "all-gathers GB", round(st[25] / 1e9, 2), "of them waited for where issued",
                  round(st[39] / 1e9, 2))
            # (round 5) the far-row gathers of all but the last block column of an outer block run beside the chain: at most
            # a fifth of the gathered volume is waited for where it is issued (measured: 16.6 % at 8 ranks, 16.5 % at 2)
            assert 0 < st[39] <= 0.2 * st[25], (world, st[25], st[39])
        assert loads.max() <= 1.05 * loads.mean() and loads.min() >= 0.95 * loads.mean(), loads / loads.mean()
        assert int(shared.sum()) <= 64                  # a handful of shared fronts, the rest private subtrees
    S.free_factor(Lf); S.free_sparse(A); S.finish()


@pytest.mark.parametrize("world", [2, 5, 8])
def test_partition_without_etree_postorder(world):
    """Common->postorder = FALSE with a random UserPerm: supernodes are
    not numbered in etree postorder, a subtree is not an index range.  Groups must
    still nest (a child's group inside its parent's) and solo subtrees stay whole."""
    n, Ap, Ai, Ax = G.poisson2d(30)
    perm = np.random.default_rng(30).permutation(n)
    owners, _, (sparent, level), _, _ = _host_plans(world, n, Ap, Ai, Ax, perm, postorder=False)
    o = owners[0]
    for other in owners[1:]:
        assert np.array_equal(o, other)
    g0, gn = _host_plans.groups[0]
    hp = np.where(sparent >= 0)[0]
    pa = sparent[hp]
    assert np.all((g0[hp] >= g0[pa]) & (g0[hp] + gn[hp] <= g0[pa] + gn[pa]))
    shared = o < 0
    assert np.all(shared[sparent[shared & (sparent >= 0)]])
    solo = ~shared & (sparent >= 0)
    assert np.all((o[sparent[solo]] == o[solo]) | (o[sparent[solo]] < 0))
    assert np.all(gn[shared] >= 2) and np.all(gn[~shared] == 1)


def test_world_one_has_no_shared_fronts():
    n, Ap, Ai, Ax = G.poisson3d(10)
    owners, _, _, _, _ = _host_plans(1, n, Ap, Ai, Ax, G.geometric_nd(10, 10, 10, 3))
    assert np.all(owners[0] == 0)


@pytest.mark.parametrize("world", [2, 3])
def test_allreduce_callback_over_gloo_cpu(world):
    res = _run_ranks(world, "cpu", "-")
    assert all(r["ok"] and r["calls"] == 1 for r in res)


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [(2, "p3d_20"), (3, "p3d_32"), (2, "p2d_90"), (4, "box10"),
                                        (4, "p3d_32"), (2, "p3d_48")])
def test_distributed_factorization_matches_oracle(world, case):
    res = _run_ranks(world, "gpu", case)
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0, r
        assert r["err"] < 1e-12, r
        assert r["resid"] < 1e-11, r
        assert r["nshared"] > 0 and r["nshared"] + sum(r["owned"]) == r["nsuper"], r
        assert r["allreduce_calls"] > 0
    assert len({json.dumps(r["owned"]) for r in res}) == 1


@pytest.mark.gpu
def test_distributed_not_posdef_protocol():
    res = _run_ranks(2, "gpu", "p3d_16_notposdef")
    for r in res:
        assert r["oracle_status"] == 1 and r["ok"] == 1 and r["status"] == ch.NOT_POSDEF, r
        assert r["minor"] == r["oracle_minor"], r
        assert r["zero_pattern_equal"] and r["err"] < 1e-12, r


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_refactorizations_clear_only_their_slabs(world):
    """Factor left on the devices: later factorizations clear only the parts of Lx a rank
    holds; after a gather (the other ranks' columns are now in its Lx) everything again."""
    res = _run_ranks(world, "gpu", "p3d_32", extra_env={"DIST_TEST_RESIDENT": "1"})
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
        assert all(e < 1e-12 for e in r["resident_errs"]), r


@pytest.mark.gpu
def test_distributed_with_memory_split_schedule():
    """Ranks + the memory-aware subtree sweep together (the 200^3 multi-GPU shape)."""
    res = _run_ranks(3, "gpu", "p3d_32", extra_env={"CHOLMOD_HIP_ARENA_BUDGET_MB": "20"})
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
        assert r["nsplit"] > 1, r


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [128, 128 | 256, 256])
def test_distributed_with_wide_outer_blocks(flags):
    """Mid-level (K = 512) updates of a wide outer block are dealt to the ranks
    and every 512-column block is summed before it is factored -- ahead of time,
    overlapped with the rest of the trailing update, unless flag 256
    (CHOLMOD_HIP_NO_EXCHANGE_LOOKAHEAD) asks for the plain order."""
    res = _run_ranks(3, "gpu", "p3d_32", extra_env={"CHOLMOD_TEST_HIP_FLAGS": str(flags)})
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
        assert r["allreduce_calls"] >= 3, r


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["p3d_32", "p3d_16_notposdef"])
def test_exchange_path_over_rccl_single_rank(case):
    """The 1-GPU box cannot host two RCCL ranks, so the engine's self test
    CHOLMOD_HIP_SHARE_AS_WORLD marks the fronts a 4-rank run would share and the
    whole exchange path (slab packing, zero-copy tensor view, torch.distributed
    "nccl" all-reduce, unpack, not-posdef agreement) runs with one rank."""
    res = _run_ranks(1, "gpu", case, extra_env={"DIST_TEST_BACKEND": "nccl",
                                                  "CHOLMOD_HIP_SHARE_AS_WORLD": "4"})
    r = res[0]
    assert r["nshared"] > 0 and r["allreduce_calls"] > r["nshared"] // 2, r
    assert r["err"] < 1e-12 and r["ok"] == 1, r
    if case.endswith("notposdef"):
        assert r["status"] == ch.NOT_POSDEF and r["minor"] == r["oracle_minor"], r
    else:
        assert r["status"] == 0 and r["resid"] < 1e-11, r


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead", [True, False])
@pytest.mark.parametrize("notposdef", [False, True])
def test_native_rccl_exchange_single_rank(monkeypatch, lookahead, notposdef):
    """The engine's own RCCL path (cholmod_hip_rccl_unique_id / _attach: dlopen'ed
    librccl, ncclCommInitRank, stream-ordered ncclAllReduce on the engine's
    streams, no host callback, no torch) with the one rank a 1-GPU box can host:
    CHOLMOD_HIP_SHARE_AS_WORLD marks the fronts a 4-rank run would share, so packing,
    all-reduce, unpack, the exchange look-ahead on the second stream and the
    not-posdef agreement all run; the factor must match the oracle."""
    from oracle.oracle import OracleFactor
    monkeypatch.setenv("CHOLMOD_HIP_SHARE_AS_WORLD", "4")
    n, Ap, Ai, Ax = G.poisson3d(20)
    perm = G.geometric_nd(20, 20, 20, 4)
    O = OracleFactor(n, Ap, Ai, -1, perm=perm, postorder=True)
    if notposdef:
        kbad = int(O.super[O.nsuper - 3] + 2)
        Ax = Ax.copy()
        Ax[Ap[int(O.Perm[kbad])]] = -5.0
        assert O.factorize(Ax) == 1 and O.minor == kbad
    else:
        assert O.factorize(Ax) == 0
    S = ch.Session(hip_flags=0 if lookahead else 256)
    A = S.sparse(n, Ap, Ai, Ax, -1)
    Lf = S.analyze(A, perm)
    assert S.L.cholmod_l_hip_prepare(Lf, C.byref(S.cm)) == 1
    plan = ch.FactorView(Lf).hip_plan
    idb = np.zeros(128, dtype=np.uint8)
    assert S.L.cholmod_hip_rccl_unique_id(idb.ctypes.data) == 0
    assert S.L.cholmod_hip_rccl_attach(plan, idb.ctypes.data) == 0
    assert S.factorize(A, Lf) == 1
    st = S.hip_stats(Lf)
    assert st[17] > 0 and st[18] > 0                     # all-reduce launches really in the schedule
    fv = ch.FactorView(Lf)
    m = O.lower_mask()
    assert np.linalg.norm((fv.x - O.x)[m]) <= 1e-12 * np.linalg.norm(O.x[m])
    if notposdef:
        assert S.cm.status == ch.NOT_POSDEF and fv.minor == O.minor
    else:
        assert S.cm.status == ch.OK
        b = G.demo_rhs(n)
        x = S.solve(Lf, b)
        assert np.linalg.norm(G.sym_matvec(n, Ap, Ai, Ax, -1, x) - b) <= 1e-11 * np.linalg.norm(b)
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()


STANDIN = os.path.join(ROOT, "tests", "standin_rccl", "libstandin_rccl.so")
# (GPU_MAX_HW_QUEUES: the stand-in's collectives are asynchronous -- a small kernel on the caller's stream waits for the
# helper thread's flag; streams that shared a hardware queue with it would wait behind it)
NATIVE = {"DIST_TEST_EXCHANGE": "native", "CHOLMOD_HIP_RCCL_LIBRARY": STANDIN, "GPU_MAX_HW_QUEUES": "8"}


def test_standin_collective_library_exports_what_the_engine_binds():
    """tests/standin_rccl (built by __graft_entry__.build()): the nccl* entry points
    engine.hip: rccl_api() binds by name."""
    lib = C.CDLL(STANDIN)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommSplit", "ncclAllReduce", "ncclReduceScatter",
                 "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(lib, name), name


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [(2, "p3d_20"), (3, "p3d_32"), (2, "p2d_90"), (4, "box10"),
                                        (4, "p3d_32"), (2, "p3d_48")])
def test_native_exchange_between_ranks_matches_oracle(world, case):
    """The engine-native exchange with REAL peers (round-2 review, item 1): 2-4 processes share
    GPU 0, each attaches its plan with cholmod_hip_rccl_attach -- ncclCommInitRank, one
    ncclCommSplit per rank group, reduce-scatter of every shared block column by row chunks,
    panel chain on the own rows, all-gather, the final agreement all-reduce, the gather of the
    factor -- over the stand-in collective library (CHOLMOD_HIP_RCCL_LIBRARY), and checks its
    gathered factor against the oracle."""
    res = _run_ranks(world, "gpu", case, extra_env=NATIVE)
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0, r
        assert r["err"] < 1e-12, r
        assert r["resid"] < 1e-11, r
        assert r["nshared"] > 0 and r["nshared"] + sum(r["owned"]) == r["nsuper"], r
        assert r["allreduce_calls"] > 0
    assert len({json.dumps(r["owned"]) for r in res}) == 1
    # cholmod_hip_factor_checks_local: the ranks' shares of the invariants (log det / 2, dead-triangle entries,
    # non-finite entries, ||L||_F^2, non-positive pivots) add up to those of the gathered factor
    tot = np.sum([r["checks_local"] for r in res], axis=0)
    full = np.array(res[0]["checks_full"])
    assert abs(tot[0] - full[0]) <= 1e-11 * abs(full[0]) and abs(tot[3] - full[3]) <= 1e-11 * full[3], (tot, full)
    assert tot[1] == full[1] == 0 and tot[2] == full[2] == 0 and tot[4] == full[4] == 0, (tot, full)


# ---- the headline world size with real peers (round-4 review, item 1) -----------------------------------------------------
# Eight processes share GPU 0 and drive the engine-native exchange through the asynchronous stand-in: the nested 8 / 4 / 2
# group structure an 8-rank partition produces (sub-communicators by ncclCommSplit, collectives of different groups
# interleaved on every rank), streams jittered, arena / windows / staging poisoned.

JIT8 = dict(CHOLMOD_HIP_TEST_POISON_ARENA="1", CHOLMOD_HIP_TEST_JITTER="11:1200")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["p3d_40", pytest.param("p3d_64", marks=pytest.mark.slow)])
def test_world8_on_one_gpu_matches_oracle(case):
    """(40^3 in the default -m gpu set; 64^3 -- eight oracle factorizations beside the eight peers, 100 s -- under
    -m "gpu and slow")"""
    res = _run_ranks(8, "gpu", case, timeout=1500, extra_env=dict(NATIVE, **JIT8))
    sizes = set()
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0, r
        assert r["err"] < 1e-12 and r["resid"] < 1e-11, r
        assert r["nshared"] > 0 and r["nshared"] + sum(r["owned"]) == r["nsuper"], r
        assert r["allreduce_calls"] > 0
        sizes.update(r["allreduce_group_sizes"])
    assert 8 in sizes and (4 in sizes or 2 in sizes), sizes         # groups smaller than the world took part
    assert len({json.dumps(r["owned"]) for r in res}) == 1
    tot = np.sum([r["checks_local"] for r in res], axis=0)
    full = np.array(res[0]["checks_full"])
    assert abs(tot[0] - full[0]) <= 1e-11 * abs(full[0]) and abs(tot[3] - full[3]) <= 1e-11 * full[3], (tot, full)
    assert tot[1] == full[1] == 0 and tot[2] == full[2] == 0 and tot[4] == full[4] == 0, (tot, full)


@pytest.mark.gpu
@pytest.mark.slow
def test_world8_on_one_gpu_at_100_cubed_distributed_checks():
    """BASELINE configs[1] (Poisson 100^3, 1 M dof) on eight peers: the factor stays distributed (1 / 8 of L + windows per
    rank), its invariants are summed over the ranks and held against the closed forms (log det A, trace A), first
    factorization and resident refactorization."""
    res = _run_ranks(8, "gpu", "checks_p3d_100", timeout=1500, extra_env=dict(NATIVE, **JIT8))
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0, r
        for c in r["checks"]:
            assert c["logdet_rel_err"] < 1e-11 and c["trace_rel_err"] < 1e-11, c
            assert c["upper_nonzeros"] == 0 and c["nonfinite"] == 0 and c["nonpositive_diag"] == 0, c
        assert r["exchanges"] > 20 and r["L_bytes_rank"] < 0.4 * r["L_bytes_whole"], r


@pytest.mark.gpu
def test_world8_not_posdef_agreement():
    """A failing pivot in the middle of the root (shared by all eight ranks, past its first 64-column step): every rank
    returns the same minor and the oracle's zero pattern."""
    res = _run_ranks(8, "gpu", "p3d_32_notposdef_root", timeout=900, extra_env=dict(NATIVE, **JIT8))
    for r in res:
        assert r["oracle_status"] == 1 and r["ok"] == 1 and r["status"] == ch.NOT_POSDEF, r
        assert r["minor"] == r["oracle_minor"], r
        assert r["zero_pattern_equal"] and r["err"] < 1e-12, r


@pytest.mark.gpu
@pytest.mark.parametrize("world,flags", [(2, 64), (3, 64 | 256), (4, 0)])
def test_distributed_dense_root_windows(world, flags):
    """One dense supernode of 1400 columns shared by every rank: the panel is stored by 128-column slabs on their
    owners (a rank holds about 1 / world of it), the group factors it through windows of one outer block column
    (512 wide with flag 64: three of them, two buffers used alternately, opened ahead of time with the exchange
    look-ahead and in line without, flag 256).  From the second outer block on the window's virtual base is below
    the rank's array (negative offset): both signs must occur."""
    res = _run_ranks(world, "gpu", "dense_1400", extra_env=dict(NATIVE, CHOLMOD_TEST_HIP_FLAGS=str(flags)))
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
        assert r["nshared"] == 1 and r["nsuper"] == 1, r
        assert r["window_opens"] >= 3 and 0 < r["window_opens_negative_base"] < r["window_opens"], r
        # the rank's part: its slabs + two windows, not the whole factor (1400 x 1400)
        slabs = sum(min(128, 1400 - c0) for c0 in range(0, 1400, 128) if (c0 // 128) % world == r["rank"])
        wins = 2 * 1400 * 512           # (a front of 1400 rows has 512-column outer blocks with or without flag 64)
        assert r["L_bytes_rank"] == 8.0 * (slabs * 1400 + wins), r


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_complex_input(world):
    """Hermitian input on several ranks (reference templates t_cholmod_super_numeric.c:41-83): the engine factors the
    real twin with the even-column update kernels, shared fronts distributed by slabs (a slab boundary is an even
    column), and every rank gathers the complex factor: against the oracle's zherk / zpotrf restatement."""
    res = _run_ranks(world, "gpu", "p3d_20_complex")
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11 and r["upper_zero"], r


@pytest.mark.gpu
@pytest.mark.parametrize("own_w", [64, 256, 512])
def test_distributed_slab_widths(own_w):
    """The slab width of the distributed fronts (default 128): one slab per tile column, half a block column, a whole one."""
    res = _run_ranks(3, "gpu", "p3d_32", extra_env=dict(NATIVE, CHOLMOD_HIP_OWN_W=str(own_w), CHOLMOD_HIP_TEST_POISON_ARENA="1"))
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
    assert len({r["L_bytes_rank"] for r in res}) > 1 or own_w == 512, res      # (the ranks hold different slabs)


@pytest.mark.gpu
@pytest.mark.parametrize("env,flags", [({"CHOLMOD_HIP_SHARED_CHAIN64": "1"}, 0), pytest.param({"CHOLMOD_HIP_NO_CHAINF": "1"}, 8192, marks=pytest.mark.slow),
                                       pytest.param({"CHOLMOD_HIP_NARROW_EXCHANGE_KERNELS": "1"}, 0, marks=pytest.mark.slow),
                                       ({"CHOLMOD_HIP_NO_CB_PASSTHROUGH": "1"}, 0),
                                       pytest.param({"CHOLMOD_HIP_NO_CB_PASSTHROUGH": "1", "CHOLMOD_HIP_NO_CB_BALANCE": "1"}, 256,
                                                    marks=pytest.mark.slow)])
def test_distributed_chain_variants(env, flags):
    """Shared fronts take the fused 256-column chain (k_chainf) by default; the other forms stay under test: the 64-column
    chain on shared fronts, the two-kernel 256-column chain (k_diag + k_rowsolve) through the windows, the exchange
    stream's kernels as one-wave workgroups, and the contribution blocks of shared fronts as full squares of partial sums
    pulled level by level (the default routes the contributions past them, CHOLMOD_HIP_NO_CB_PASSTHROUGH)."""
    res = _run_ranks(3, "gpu", "p3d_48", extra_env=dict(NATIVE, CHOLMOD_TEST_HIP_FLAGS=str(flags), CHOLMOD_HIP_TEST_POISON_ARENA="1", **env))
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r


@pytest.mark.gpu
@pytest.mark.parametrize("world,case,seed", [(4, "p3d_32", 1), (3, "p3d_48", 2), (2, "dense_1400", 3)])
def test_distributed_under_stream_jitter(world, case, seed):
    """CHOLMOD_HIP_TEST_JITTER: ahead of one launch in three its stream is held up for a random time (tens of microseconds,
    now and then up to 1.5 ms), differently on every rank -- a launch that reads what the other stream (or a collective)
    produces without an event in between then runs before its producer, and with the arena, the windows and the staging
    buffers poisoned that shows in the factor."""
    res = _run_ranks(world, "gpu", case, extra_env=dict(NATIVE, CHOLMOD_HIP_TEST_POISON_ARENA="1", CHOLMOD_HIP_TEST_JITTER=f"{seed}:1500"))
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r


@pytest.mark.gpu
def test_distributed_jitter_catches_dropped_waits():
    """The mutation the jitter must catch: the same run with the schedule's cross-stream waits skipped
    (CHOLMOD_HIP_TEST_DROP_WAITS) does not reproduce the oracle's factor."""
    try:
        res = _run_ranks(3, "gpu", "p3d_48", extra_env=dict(NATIVE, CHOLMOD_HIP_TEST_POISON_ARENA="1", CHOLMOD_HIP_TEST_JITTER="5:1500",
                                                             CHOLMOD_HIP_TEST_DROP_WAITS="1"))
    except (AssertionError, RuntimeError, subprocess.SubprocessError):
        return                      # (a rank that reads a poisoned buffer too early may just as well die: noticed all the same)
    assert not all(r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 for r in res), res


# outer block columns of 2048 columns at this size: four 512-column block columns each, so that the far-row gathers of the
# first three run on the exchange stream beside the chain of the following ones (schedule_dense.hip: emit_ag)
WIDE_OUTER = {"CHOLMOD_HIP_OB1024_ROWS": "300", "CHOLMOD_HIP_OB2048_ROWS": "900"}


@pytest.mark.gpu
def test_far_row_gathers_run_beside_the_chain():
    """Review item 6 of round 4: the all-gather of a block column is split -- the near rows (inside the outer block column)
    in line, the far rows on the exchange stream, met by the main stream ahead of the outer update.  Three peers, streams
    jittered, arena / windows / staging poisoned: the factor is the oracle's, and most of the gathered volume is NOT waited
    for where it is issued."""
    res = _run_ranks(3, "gpu", "p3d_48", extra_env=dict(NATIVE, CHOLMOD_HIP_TEST_POISON_ARENA="1", CHOLMOD_HIP_TEST_JITTER="7:1500",
                                                         **WIDE_OUTER))
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
        assert 0 < r["gather_inline_MB"] < 0.6 * r["gather_MB"], r


@pytest.mark.gpu
def test_jitter_catches_a_dropped_join_with_the_far_row_gathers():
    """... and the mutation that gives it teeth: with only the joins skipped (CHOLMOD_HIP_TEST_DROP_WAITS=2) the outer
    update reads far rows nobody has gathered yet."""
    try:
        res = _run_ranks(3, "gpu", "p3d_48", extra_env=dict(NATIVE, CHOLMOD_HIP_TEST_POISON_ARENA="1", CHOLMOD_HIP_TEST_JITTER="7:1500",
                                                             CHOLMOD_HIP_TEST_DROP_WAITS="2", **WIDE_OUTER))
    except (AssertionError, RuntimeError, subprocess.SubprocessError):
        return
    assert not all(r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 for r in res), res


@pytest.mark.gpu
def test_distributed_not_posdef_inside_a_shared_front():
    """The failing pivot lies in the root, a front shared by all ranks and factored through its windows: same minor, same
    zero pattern as the reference's repeat-supernode pass on every rank."""
    res = _run_ranks(3, "gpu", "p3d_16_notposdef_root", extra_env=NATIVE)
    for r in res:
        assert r["oracle_status"] == 1 and r["ok"] == 1 and r["status"] == ch.NOT_POSDEF, r
        assert r["minor"] == r["oracle_minor"] and r["zero_pattern_equal"] and r["err"] < 1e-12, r


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 256, 128 | 256])
def test_native_exchange_both_orders_and_subgroups(flags):
    """World of 4 with the heavy children of the root in sub-groups of 2 (communicator
    splits), exchange look-ahead on (second stream, event-ordered) and off (flag 256), wide
    outer blocks (flag 128)."""
    res = _run_ranks(4, "gpu", "p3d_32", extra_env=dict(NATIVE, CHOLMOD_HIP_SPLIT_TOL="1.5",
                                                         CHOLMOD_TEST_HIP_FLAGS=str(flags)))
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
    sizes = set()
    for r in res:
        sizes.update(r["allreduce_group_sizes"])
    assert 4 in sizes and (2 in sizes or 3 in sizes), sizes


@pytest.mark.gpu
@pytest.mark.parametrize("world,native", [pytest.param(3, True, marks=pytest.mark.slow), (4, True), (2, False)])
def test_distributed_with_wave_tile_kernel_on_dealt_tiles(world, native):
    """The update kernel an 8-GPU run of the headline spends its time in (k_update3, one wave per tile) on
    the tiles of shared fronts that are DEALT over the ranks of a group (GemmGroup.tile_mul / tile_add):
    the small test problems never reach its 2048-tile threshold, so CHOLMOD_HIP_UPD3_MIN_TILES=1 sends
    every region through it."""
    env = dict(NATIVE if native else {}, CHOLMOD_HIP_UPD3_MIN_TILES="1")
    res = _run_ranks(world, "gpu", "p3d_48" if world < 4 else "p3d_32", extra_env=env)
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r


@pytest.mark.gpu
def test_world4_on_one_gpu_repeatedly():
    """Four processes time-slicing one GPU, five factorizations in a row, every dense update through the
    one-wave-per-tile kernel: the run that exposed the diagonal-row race of k_thin_front<4> in round 3 (a wave held
    back between two of its loads read diagonal rows wave 0 had already finished: the last columns of that wave's
    rows off by a wrong pivot, one run in three; tf_panel now waits until every wave has its copy)."""
    env = dict(NATIVE, CHOLMOD_HIP_UPD3_MIN_TILES="1", CHOLMOD_HIP_TEST_POISON_ARENA="1")
    for _ in range(5):
        res = _run_ranks(4, "gpu", "p3d_32", extra_env=env)
        for r in res:
            assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r


@pytest.mark.gpu
def test_native_exchange_not_posdef_agreement():
    res = _run_ranks(3, "gpu", "p3d_16_notposdef", extra_env=NATIVE)
    for r in res:
        assert r["oracle_status"] == 1 and r["ok"] == 1 and r["status"] == ch.NOT_POSDEF, r
        assert r["minor"] == r["oracle_minor"], r
        assert r["zero_pattern_equal"] and r["err"] < 1e-12, r


@pytest.mark.gpu
def test_native_exchange_with_memory_split_and_resident_refactorizations():
    res = _run_ranks(3, "gpu", "p3d_32", extra_env=dict(NATIVE, CHOLMOD_HIP_ARENA_BUDGET_MB="20", DIST_TEST_RESIDENT="1"))
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
        assert all(e < 1e-12 for e in r["resident_errs"]), r
        assert r["nsplit"] > 1, r


@pytest.mark.gpu
def test_native_exchange_local_failure_does_not_hang_the_other_ranks():
    res = _run_ranks(3, "gpu", "p3d_20", timeout=300, extra_env=dict(NATIVE, CHOLMOD_HIP_TEST_FAIL_LAUNCH="1:25"))
    for r in res:
        assert r["ok"] == 0 and r["status"] == ch.GPU_PROBLEM, r


@pytest.mark.gpu
@pytest.mark.parametrize("native", [True, False])
def test_a_rank_without_room_for_the_gathered_factor_does_not_hang_the_others(native):
    """cholmod_hip_gather_factor builds the complete factor on every rank (181.6 GB at the headline size, next to
    the rank's own 117 GB at two ranks: more than one MI355X holds).  A rank that finds no room (test hook) tells
    the others before anybody enters the sums: every rank returns CHOLMOD_OUT_OF_MEMORY, and the factor -- still
    distributed -- gathers fine once there is room."""
    res = _run_ranks(3, "gpu", "p3d_20", timeout=300, extra_env=dict(NATIVE if native else {}, CHOLMOD_HIP_TEST_FAIL_GATHER="1",
                                                                      DIST_TEST_RESIDENT_PLAIN="1"))
    for r in res:
        assert r["gather_failed"] == 0 and r["gather_failed_status"] == ch.OUT_OF_MEMORY, r
        assert r["gather_again"] == 1 and r["err"] < 1e-12, r


@pytest.mark.gpu
@pytest.mark.parametrize("native", [True, False])
def test_gather_through_host_memory_when_the_factor_does_not_fit_next_to_the_own_part(native):
    """Rank 1 (test hook) finds no room for the complete factor next to its own part: it downloads its part, releases
    it, reserves the complete array and uploads its fronts into place -- the others gather directly.  The gathered factor
    and a solve are right; so are the next factorization (the rank reserves its own array again) and its gather."""
    res = _run_ranks(3, "gpu", "p3d_20", timeout=300, extra_env=dict(NATIVE if native else {}, CHOLMOD_HIP_TEST_GATHER_STAGED="1",
                                                                      DIST_TEST_RESIDENT_PLAIN="1"))
    for r in res:
        e = r["staged_errs"]
        assert e[0] < 1e-12 and e[2] < 1e-12 and e[1] < 1e-11 and e[3] < 1e-11, r


@pytest.mark.gpu
def test_local_failure_does_not_hang_the_other_ranks():
    """A launch of rank 1 fails in the middle of the factorization (test hook): rank 1
    keeps taking part in the remaining collectives and the failure travels through
    the final agreement exchange, so every rank returns CHOLMOD_GPU_PROBLEM."""
    res = _run_ranks(3, "gpu", "p3d_20", timeout=300, extra_env={"CHOLMOD_HIP_TEST_FAIL_LAUNCH": "1:25"})
    for r in res:
        assert r["ok"] == 0 and r["status"] == ch.GPU_PROBLEM, r


@pytest.mark.gpu
def test_distributed_rank_subgroups():
    """Proportional mapping: with 4 ranks the heavy children of the root split its
    group, so block columns are summed over 2-rank sub-groups as well as over
    everybody -- and the factor still matches the oracle on every rank."""
    res = _run_ranks(4, "gpu", "p3d_32", extra_env={"CHOLMOD_HIP_SPLIT_TOL": "1.5"})
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
    sizes = set()
    for r in res:
        sizes.update(r["allreduce_group_sizes"])
    assert 4 in sizes and (2 in sizes or 3 in sizes), sizes


@pytest.mark.gpu
def test_distributed_with_mixed_outer_block_widths():
    """Outer block width is per front: with the row thresholds lowered, the root
    (1609 rows) is cut in 2048-wide, its children in 1024-wide and the rest in
    512-wide outer block columns inside the same batches, on every rank alike."""
    res = _run_ranks(4, "gpu", "p3d_32", extra_env={"CHOLMOD_HIP_OB1024_ROWS": "500",
                                                     "CHOLMOD_HIP_OB2048_ROWS": "1200"})
    for r in res:
        assert r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 and r["resid"] < 1e-11, r
