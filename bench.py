#!/usr/bin/env python3
"""bench.py -- factorize GFLOP/s of the supernodal Cholesky hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line from rank 0.  A "step" is one numeric factorization
(cholmod_l_factorize's device part: assemble + all fronts) of the workload with
the permuted input matrix already resident in HBM.  Metric = Common->fl / t,
fl = sum_j ColCount[j]^2 (reference CHOLMOD/Cholesky/cholmod_rowcolcounts.c:
517-528, demo convention CHOLMOD/Demo/cholmod_l_demo.c:691-692).

Workload: the configuration BASELINE.json's metric is quoted on, 3D 7-point
Poisson 200^3 (8M dof, fl = 4.25e14) under geometric nested dissection (SURVEY.md
8d) -- it fits one MI355X (L 181.6 GB + contribution blocks).  If the
device cannot hold it the bench falls back to 160^3, then 100^3 (configs[1]), and
names the workload it actually ran.  N>1: the SAME factorization partitioned over
the ranks (one process per GPU): private etree subtrees per rank, the panels of the
shared top fronts distributed by column slabs (owner-computes), each 512-column block
column summed with a reduce-scatter by row chunks before its panel chain and gathered
after it -- the engine calls RCCL itself; strong scaling.
`--gpus N` without a launcher around it starts the N ranks itself.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6        # MI355X spec, dense fp64 matrix


def build_workload(name, m):
    from suitesparse_amd import generators as G
    if name == "poisson3d":
        n, Ap, Ai, Ax = G.poisson3d(m)
        return n, Ap, Ai, Ax, -1, G.geometric_nd(m, m, m, 4), f"poisson3d_{m}^3_geometricND_leaf4"
    if name == "poisson2d":
        n, Ap, Ai, Ax = G.poisson2d(m)
        return (n, Ap, Ai, Ax, -1, G.geometric_nd(m, m, 1, 4),
                f"poisson2d_{m}^2_geometricND_leaf4 (G3_circuit stand-in, SURVEY 8d)")
    if name == "box3d":
        n, Ap, Ai, Ax = G.box_stencil3d(m, 3)
        return (n, Ap, Ai, Ax, -1, G.geometric_nd(m, m, m, 6, 3),
                f"box_stencil_r3_{m}^3_geometricND_leaf6_width3 " + ("(nd24k stand-in, SURVEY 8d)" if m == 42 else "(nd24k stand-in's stencil, flop-matched grid)"))
    raise ValueError(name)


def _scipy_openblas():
    """LP64 OpenBLAS shipped with scipy, in CHOLMOD_BLAS_LIBRARY syntax (path:symbol-prefix), or None."""
    import glob
    try:
        import scipy
        d = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
        c = sorted(glob.glob(os.path.join(d, "libscipy_openblas-*.so")))
        return (c[0] + ":scipy_") if c else None
    except Exception:
        return None


CPU_CHILD = r"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.environ["BENCH_ROOT"])
from suitesparse_amd import cholmod as ch, generators as G
m = int(os.environ["BENCH_CPU_M"])
counts = [int(v) for v in os.environ["BENCH_CPU_COUNTS"].split(",")]
L0 = ch.lib()
cap = int(L0.ssamd_cpu_max_threads())          # (OpenBLAS: its compile-time MAX_THREADS; 0 = none)
if cap > 0:
    counts = sorted({min(c, cap) for c in counts})
budget = float(os.environ.get("BENCH_CPU_BUDGET_S", "18"))
gomp = ctypes.CDLL("libgomp.so.1")
n, Ap, Ai, Ax = G.poisson3d(m)
perm = G.geometric_nd(m, m, m, 4)
S = ch.Session(use_gpu=0)
A = S.sparse(n, Ap, Ai, Ax, -1)
Lf = S.analyze(A, perm)
# one untimed factorization on all threads: the first touch of L->x (10.5 GB at 100^3) is page faults, not arithmetic
t0 = time.perf_counter()
ok = S.factorize(A, Lf)
first = time.perf_counter() - t0
assert ok == 1 and S.cm.status == 0
pts, spent = [], 0.0
order = [int(v) for v in os.environ.get("BENCH_CPU_ORDER", "").split(",") if v] or sorted(counts, reverse=True)
for t in [c for c in order if c in counts]:      # the usable width first: should the budget run out, the side points are the ones missing
    if spent > budget:
        break
    gomp.omp_set_num_threads(t)
    t0 = time.perf_counter()
    ok = S.factorize(A, Lf)
    dt = time.perf_counter() - t0
    assert ok == 1 and S.cm.status == 0
    spent += dt
    pts.append({"threads": t, "seconds": dt})
S.L.ssamd_cpu_blas_name.restype = ctypes.c_char_p
print(json.dumps({"fl": S.cm.fl, "points": pts, "first_seconds": first, "blas": S.L.ssamd_cpu_blas_name().decode(), "blas_max_threads": cap}))
"""


def cpu_quota():
    """CPUs' worth of time this container may use (cgroup v2 cpu.max / v1 cfs quota), or None: the GPU boxes show 256
    hardware threads and a quota of 16 -- threads beyond it are throttled, not added."""
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if a == "max" else max(1, -(-int(a) // int(b)))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return max(1, -(-q // p_)) if q > 0 and p_ > 0 else None
    except Exception:
        return None


def cpu_baseline(sample_m):
    """The build's own CPU supernodal path (Common->useGPU = 0: suitesparse_amd/csrc/host/cpu_numeric.c, the reference's
    left-looking loop with a BLAS bound at run time -- SURVEY 8d's "the build's CPU supernodal path") timed on the host
    cores on a bounded sample of the same workload family, in ONE child process (the BLAS binding is per process).  Not the
    oracle: nothing under oracle/ is timed.  The sample is BASELINE configs[1] itself (Poisson 100^3, fl = 6.3e12); after
    one untimed factorization (first touch of L->x) the child times one factorization per thread count -- what the container
    may use (its cgroup CPU quota: 16 on the GPU boxes, whose 256 hardware threads are `host_cores`), half of that, and twice
    the quota if 18 s have not been spent by then; `value` is the best, `cores` the thread count that gave it.  Round 6: the path runs independent subtrees on one thread each and the top
    supernodes by tiles, the BLAS on one thread per call (rounds 3-5: a threaded BLAS under a serial loop, best at 16
    threads and slower beyond)."""
    import subprocess
    cores = os.cpu_count() or 1
    base = dict(os.environ, BENCH_ROOT=ROOT, BENCH_CPU_M=str(sample_m))
    if "CHOLMOD_BLAS_LIBRARY" not in base:
        b = _scipy_openblas()
        if b:
            base["CHOLMOD_BLAS_LIBRARY"] = b
    quota = cpu_quota()
    usable = min(cores, quota) if quota else cores
    if "OMP_NUM_THREADS" in os.environ:
        counts = [int(os.environ["OMP_NUM_THREADS"])]
    else:
        # up to what the container may use: every hardware thread without a quota, the quota's worth of threads with one
        # (and one point at twice the quota, to show what oversubscribing it costs)
        counts = sorted({max(1, usable // 2), usable} | ({min(cores, 2 * usable)} if quota else set()))
    base["BENCH_CPU_COUNTS"] = ",".join(str(c) for c in counts)
    # (what the container may use first, half of it next, twice the quota -- what oversubscribing it costs -- if time is left)
    base["BENCH_CPU_ORDER"] = ",".join(str(c) for c in ([usable, max(1, usable // 2)] + ([min(cores, 2 * usable)] if quota else [])))
    env = dict(base, OMP_NUM_THREADS=str(max(counts)), OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", OMP_PROC_BIND="false")
    t0 = time.perf_counter()
    try:
        out = subprocess.run([sys.executable, "-c", CPU_CHILD], env=env, capture_output=True, text=True, timeout=900)
        r = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:      # report, never fail the bench line
        tail = ""
        try:
            tail = " | " + " ".join((out.stderr or "").strip().splitlines()[-3:])
        except Exception:
            pass
        return {"value": None, "unit": "GFLOP/s", "cores": 0, "host_cores": cores, "kind": "port", "sample": (f"failed: {e!r}" + tail)[:300]}
    leg = time.perf_counter() - t0
    fl, blas = r["fl"], r["blas"]
    pts = sorted(({"threads": q["threads"], "GFLOPs": fl / q["seconds"] / 1e9, "seconds_best": q["seconds"]} for q in r["points"]),
                 key=lambda q: q["threads"])
    top = max(pts, key=lambda q: q["GFLOPs"])
    return {"value": top["GFLOPs"], "unit": "GFLOP/s", "cores": top["threads"], "host_cores": cores, "cpu_quota": quota, "kind": "port",
            "sample_short": f"poisson3d {sample_m}^3 ND, whole factorization, {top['seconds_best']:.1f} s",
            "path": "product CPU path (cholmod_l_factorize with Common->useGPU = 0, host/cpu_numeric.c)",
            "by_threads": pts, "first_factorization_seconds": r["first_seconds"], "leg_seconds": leg,
            "blas_max_threads": r.get("blas_max_threads"),
            "monotone_in_threads": all(pts[i + 1]["GFLOPs"] >= 0.97 * pts[i]["GFLOPs"] for i in range(len(pts) - 1)
                                       if pts[i + 1]["threads"] <= usable),
            "sample": f"poisson3d {sample_m}^3 geometric ND" + (" (BASELINE configs[1], the whole factorization)" if sample_m == 100 else "")
                      + f", one factorization per thread count ({', '.join(str(q['threads']) for q in pts)}) after one untimed, "
                      f"fl={fl:.3e}, {top['seconds_best']:.2f} s at {top['threads']} threads, BLAS={blas} (one thread per call), host cores {cores}"
                      + (f", container CPU quota {quota}" if quota else "")}


PMC_BY_WORKLOAD = {"poisson3d_200^3_geometricND_leaf4": "r05k_pmc_summary_poisson200_top48.json"}
# counters of EVERY launch of one refactorization of the mid-size workloads, summed per kernel (tools/pmc_workload.sh: three
# separate rocprofv3 --pmc passes; round-4 review, item 3) -- matched by the start of the workload name
PMC_BY_KERNEL = {"poisson3d_100^3": "r05k_pmc_by_kernel_p100.json", "box_stencil_r3_42^3": "r06p_pmc_by_kernel_box42r3.json",
                 "poisson2d_1259^2": "r05k_pmc_by_kernel_p2d1259.json"}
CHAIN_KERNELS = ("k_update2f", "k_trsm_upd", "k_trsm_mfma", "k_potrf_mfma", "k_extend_add", "k_update2", "k_update3", "k_thin_front",
                 "k_leaf_pair")


def pmc_by_kernel(wname):
    """(dict kernel -> counters, file name) of the committed per-kernel counter summary of this workload, or (None, None)."""
    for pre, fn in PMC_BY_KERNEL.items():
        if wname.startswith(pre):
            pf = os.path.join(ROOT, "profiles", fn)
            if os.path.exists(pf):
                return json.load(open(pf)), fn
    return None, None


def pmc_kernel_rows(pj, prefix):
    """Counters of every instantiation of a kernel (k_update3<4, 0, 1>, k_update3<2, 0, 1>, ...) added up."""
    acc = {}
    for k, v in pj.items():
        if k == prefix or k.startswith(prefix + "<"):
            for c, x in v.items():
                acc[c] = acc.get(c, 0.0) + x
    if not acc:
        return None
    act = acc.get("GRBM_GUI_ACTIVE", 0.0)
    return {"launches": int(acc.get("dispatches", 0)),
            "traffic_bytes": 2.0 * 1024.0 * acc.get("FETCH_SIZE", 0.0) + 1024.0 * acc.get("WRITE_SIZE", 0.0),
            "fetch_bytes": 2.0 * 1024.0 * acc.get("FETCH_SIZE", 0.0), "write_bytes": 1024.0 * acc.get("WRITE_SIZE", 0.0),
            "mfma_utilisation": (acc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act / 8.0 * 1024.0)) if act > 0 else None}
# counters summed per kernel over one refactorization of the thin stand-in (tools/evidence.sh PMC=1, tools/pmc_by_kernel.py)
PMC_THIN = "r04zq_pmc_by_kernel_poisson2d1259.json"


def thin_traffic():
    """HBM-side bytes of the thin-front kernels per factorization of the G3_circuit stand-in, from the committed
    counter passes (FETCH_SIZE x 2 + WRITE_SIZE, KiB; the calibration of DESIGN section 4), or None."""
    pf = os.path.join(ROOT, "profiles", PMC_THIN)
    if not os.path.exists(pf):
        return None
    pj = json.load(open(pf))
    fetch = sum(v.get("FETCH_SIZE", 0.0) for k, v in pj.items() if k.startswith(("k_thin_front", "k_leaf_pair")))
    write = sum(v.get("WRITE_SIZE", 0.0) for k, v in pj.items() if k.startswith(("k_thin_front", "k_leaf_pair")))
    if fetch <= 0 and write <= 0:
        return None
    return {"fetch_bytes": 2.0 * 1024.0 * fetch, "write_bytes": 1024.0 * write, "traffic_bytes": 2.0 * 1024.0 * fetch + 1024.0 * write,
            "source": "profiles/" + PMC_THIN}


def roofline_of(S, Lf, wname, world):
    """Roofline object of the dominant kernel of the workload: one extra, untimed factorization with
    HIP events around every launch (on the engine's own stream).  The dense-update flops of a big
    problem run in k_update3 (one wave per 64 x 64 tile, regions of >= 2048 tiles), the rest in
    k_update2 (four waves per tile); whichever took more of this factorization is reported, the
    other beside it."""
    S.set_profiling(Lf, True)
    assert S.refactorize_resident(Lf) == 1
    ps = S.hip_stats(Lf)
    S.set_profiling(Lf, False)
    lp = S.launch_profile(Lf)
    use_w = ps[32] > ps[6]
    sec, nl, fl_, by_ = (ps[32], ps[33], ps[34], ps[35]) if use_w else (ps[6], ps[7], ps[8], ps[16])
    if sec <= 0:
        return None
    kind = 12 if use_w else 5
    kname = "k_update3<4> / <2> (one wave per 64x64 tile, no LDS)" if use_w else "k_update2<64,64,16,2,false>"
    ach = fl_ / sec / 1e12
    traffic, traffic_note, traffic_detail = None, None, None
    # HBM-side bytes per launch of this kernel from separate rocprofv3 --pmc passes of the same
    # workload (offline, profiles/README.md): FETCH_SIZE x 2 + WRITE_SIZE x 1, the calibration
    # measured on known-byte kernels
    upd = np.where(lp["kind"] == kind)[0]
    pj = None
    if wname in PMC_BY_WORKLOAD and world == 1:
        pf = os.path.join(ROOT, "profiles", PMC_BY_WORKLOAD[wname])
        pj = json.load(open(pf)) if os.path.exists(pf) else None
    if pj is not None and pj.get("kernel_kind", 5) == kind and upd.size >= pj["launches_profiled"]:
        k = pj["launches_profiled"]
        top = upd[np.argsort(-lp["ms"][upd])[:k]]       # the same selection: the k longest launches
        traffic = pj["traffic_bytes_per_launch"]
        traffic_detail = {
            "launches": int(k), "selection": pj["selection"],
            "share_of_kernel_time": float(lp["ms"][top].sum() / lp["ms"][upd].sum()),
            "ms_per_launch": float(lp["ms"][top].mean()),
            "algorithmic_bytes_per_launch": float(lp["bytes"][top].mean()),
            "algorithmic_flops_per_launch": float(lp["flops"][top].mean()),
            "TFLOPs_on_these_launches": float(lp["flops"][top].sum() / (1e-3 * lp["ms"][top].sum()) / 1e12),
            "fetch_bytes_per_launch": pj["fetch_bytes_per_launch"], "write_bytes_per_launch": pj["write_bytes_per_launch"],
            "traffic_over_algorithmic": float(pj["traffic_bytes_per_launch"] / lp["bytes"][top].mean()),
            "TBps_at_the_memory_side": float(pj["traffic_bytes_per_launch"] / (1e-3 * lp["ms"][top].mean()) / 1e12),
            "source": "profiles/" + PMC_BY_WORKLOAD[wname]}
        traffic_note = ("per launch over the %d longest launches (%.0f %% of this kernel's time): a counter pass over all "
                        "launches of a 200^3 factorization does not finish" % (k, 100 * traffic_detail["share_of_kernel_time"]))
    else:
        traffic_note = "no PMC summary under profiles/ for this workload and kernel"
    mfma_util, chain = None, None
    bk, bk_file = pmc_by_kernel(wname) if world == 1 else (None, None)
    if bk is not None and traffic is None:
        # all launches of one refactorization, summed per kernel: the dominant kernel's bytes per launch beside the algorithmic
        # bytes per launch of THIS run (same launch list: the schedule is a function of the symbolic factor)
        row = pmc_kernel_rows(bk, "k_update3" if use_w else "k_update2")
        if row and row["launches"] > 0:
            traffic = row["traffic_bytes"] / row["launches"]
            mfma_util = row["mfma_utilisation"]
            traffic_detail = {
                "launches": row["launches"], "selection": "every launch of this kernel in one refactorization",
                "share_of_kernel_time": 1.0, "ms_per_launch": 1e3 * sec / max(nl, 1),
                "algorithmic_bytes_per_launch": by_ / max(nl, 1), "algorithmic_flops_per_launch": fl_ / max(nl, 1),
                "TFLOPs_on_these_launches": ach,
                "fetch_bytes_per_launch": row["fetch_bytes"] / row["launches"], "write_bytes_per_launch": row["write_bytes"] / row["launches"],
                "traffic_over_algorithmic": traffic / max(by_ / max(nl, 1), 1.0),
                "TBps_at_the_memory_side": traffic / (sec / max(nl, 1)) / 1e12,
                "launches_in_this_run": int(nl), "source": "profiles/" + bk_file}
            traffic_note = ("FETCH_SIZE x 2 + WRITE_SIZE summed over every launch of the kernel in one refactorization "
                            "(tools/pmc_workload.sh), per launch")
        # the kernels the mid-size configurations wait for: counters per kernel, times from this run's profiled pass
        chain = {}
        for kn in CHAIN_KERNELS:
            r = pmc_kernel_rows(bk, kn)
            if r:
                chain[kn] = {"launches": r["launches"], "traffic_GB": r["traffic_bytes"] / 1e9, "mfma_utilisation": r["mfma_utilisation"]}
    other = {"kernel": "k_update2<64,64,16,2,false>" if use_w else "k_update3", "seconds": ps[6] if use_w else ps[32],
             "launches": int(ps[7] if use_w else ps[33]),
             "TFLOPs": ((ps[8] / ps[6]) if use_w and ps[6] > 0 else (ps[34] / ps[32]) if (not use_w and ps[32] > 0) else 0.0) / 1e12}
    roofline_of.last_profile = lp
    return {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
            # `traffic` is a per-launch mean over the launches the counter passes covered; the algorithmic
            # bytes OF THE SAME LAUNCHES and the ratio of the two stand beside it (the mean over all
            # launches of the kernel is `algorithmic_bytes_per_launch_all_launches` below)
            "traffic_launches": traffic_detail["launches"] if traffic_detail else None,
            "traffic_algorithmic_bytes_same_launches": traffic_detail["algorithmic_bytes_per_launch"] if traffic_detail else None,
            "traffic_over_algorithmic": traffic_detail["traffic_over_algorithmic"] if traffic_detail else None,
            "traffic_TBps_at_the_memory_side": traffic_detail["TBps_at_the_memory_side"] if traffic_detail else None,
            "traffic_source": traffic_detail["source"] if traffic_detail else None,
            "mfma_utilisation": pj.get("mfma_utilisation") if (pj is not None and traffic_detail and mfma_util is None) else mfma_util,
            "counters_by_kernel": chain, "counters_by_kernel_source": ("profiles/" + bk_file) if chain else None,
            "traffic_note": traffic_note, "traffic_detail": traffic_detail,
            "algorithmic_bytes_per_launch_all_launches": by_ / max(nl, 1),
            "algorithmic_flops_per_launch": fl_ / max(nl, 1),
            "extend_add": {"algorithmic_GB": ps[10] / 1e9, "seconds_incl_zero": ps[9]},
            "thin_front_kernel": {"fronts": int(ps[21]), "algorithmic_GB": ps[20] / 1e9,
                                   "achieved_GBps": (ps[20] / ps[19] / 1e9) if ps[19] > 0 else None,
                                   "hbm_peak_GBps": 8000.0},
            "kernel": kname, "launches": int(nl),
            "avg_launch_ms": 1e3 * sec / max(nl, 1),
            "other_update_kernel": other,
            "seconds_by_class": {"update_wave_tiles": ps[32], "update64": ps[6], "update64_K_below_512": ps[23],
                                 "update64_plus_potrf_of_next_block": ps[27], "update128": ps[14], "extend_add+zero": ps[9],
                                 "potrf": ps[11], "trsm": ps[12], "trsm_plus_K64_update_plus_potrf": ps[30], "assemble": ps[13],
                                 "thin_fronts_fused": ps[19],
                                 "total_profiled": ps[0]}}


KIND_NAMES = {0: "zero", 1: "extend_add", 2: "potrf", 3: "trsm", 4: "update128", 5: "update64", 7: "reduce_scatter", 8: "thin",
              9: "update+potrf", 10: "trsm+upd+potrf", 11: "all_gather", 12: "update_w", 13: "diag256", 14: "rowsolve",
              15: "window", 16: "chain256f"}
SM_MAX = 136            # rows up to which a front runs in the thin-front kernel (descriptors.hip.h)
UPD3_STANDALONE_TFLOPS = 75.0    # k_update3 on a 16 384^2 x 4096 region (printed live as measured_update_kernel_*)


def critical_path(S, Lf, lp, ms_step):
    """Where a mid-size factorization stands against its dependency bound (round-4 review, item 2).

    critical_path_ms: the longest root-to-leaf path of the supernodal etree, priced as
        (64-column steps on the path) x (measured floor of one step of the panel chain)
      + (levels of thin fronts on the path) x (measured floor of a thin-front launch)
      + (all update flops of the factorization) / (k_update3's standalone rate)
    -- the panel chain of a front cannot start before its children are done and cannot go faster than its dependent
    64-column steps, and the updates cannot beat the update kernel; everything else (extend-add, zero-fill, assembly) is
    taken as free.  The per-step floor is measured in THIS run: the chain alternates `trsm+upd+potrf` | `trsm`,
    `update+potrf`, i.e. three launches per 128 columns, each at the shortest duration any launch of its kind had.
    launch_list_bound_ms: the launch list the engine actually issued, every launch at max (floor of its kind, its
    algorithmic work at the roofline: flops at the standalone update rate / bytes at 8 TB/s) -- what this schedule would
    take if every kernel were perfect; measured / launch_list_bound > 1.15 means the kernels lose time, measured /
    critical_path >> 1 with launch_list_bound close to measured means the schedule (level-synchronous batches) does."""
    from suitesparse_amd import cholmod as ch
    fv = ch.FactorView(Lf)
    nsuper = fv.nsuper
    sp = np.empty(nsuper, dtype=np.int64)
    lv = np.empty(nsuper, dtype=np.int64)
    S.L.cholmod_hip_get_maps(Lf.contents.hip_plan, sp.ctypes.data, lv.ctypes.data, None)
    nscol = np.diff(fv.super)
    nsrow = np.diff(fv.pi)
    thin = nsrow <= SM_MAX
    steps = np.where(thin, 0, (nscol + 63) // 64).astype(np.int64)
    # longest path by levels: children have lower indices' levels; process in increasing level
    order = np.argsort(lv, kind="stable")
    best_steps = np.zeros(nsuper, dtype=np.int64)      # steps on the heaviest path ending in s (s included)
    best_thin = np.zeros(nsuper, dtype=np.int64)
    acc_steps = np.zeros(nsuper, dtype=np.int64)       # best over children, gathered at the parent
    acc_thin = np.zeros(nsuper, dtype=np.int64)
    kinds, msl = lp["kind"], lp["ms"]

    def floor(k):
        q = (kinds == k) & (msl > 0)
        return float(msl[q].min()) if q.any() else 0.0
    f_tu, f_tr, f_uf, f_pf, f_thin = floor(10), floor(3), floor(9), floor(2), floor(8)
    # one step of the chain: (trsm+upd+potrf) then (trsm, update+potrf) per 128 columns; without the fusions potrf + trsm + update
    if f_tu > 0 and f_uf > 0:
        step_floor = (f_tu + f_tr + f_uf) / 2.0
    else:
        step_floor = f_pf + f_tr + floor(5)
    w_thin = 1.0 if f_thin > 0 else 0.0
    for s_ in order:
        a, b = acc_steps[s_], acc_thin[s_]
        best_steps[s_] = a + steps[s_]
        best_thin[s_] = b + (1 if thin[s_] else 0)
        p_ = sp[s_]
        if p_ >= 0:
            # path cost in time decides which child path the parent continues
            if (best_steps[s_] * step_floor + best_thin[s_] * f_thin * w_thin) > (acc_steps[p_] * step_floor + acc_thin[p_] * f_thin * w_thin):
                acc_steps[p_], acc_thin[p_] = best_steps[s_], best_thin[s_]
    roots = np.where(sp < 0)[0]
    tcost = best_steps[roots] * step_floor + best_thin[roots] * f_thin
    r = roots[int(np.argmax(tcost))]
    path_steps, path_thin = int(best_steps[r]), int(best_thin[r])
    ncb = (nsrow - nscol).astype(np.float64)
    c = nscol.astype(np.float64)
    upd_flops = float((ncb * ncb * c + ncb * c * c).sum() + ((c * c * c / 3.0) * (1.0 - 1.0 / np.maximum(c / 64.0, 1.0) ** 2)).sum())
    t_chain = path_steps * step_floor
    t_thin = path_thin * f_thin
    t_upd = upd_flops / (UPD3_STANDALONE_TFLOPS * 1e12) * 1e3
    cp = t_chain + t_thin + t_upd
    # the launch list at floor-or-roofline
    upd_kinds = (4, 5, 9, 12)
    bound = 0.0
    for k in set(kinds.tolist()):
        q = kinds == k
        fk = floor(k)
        if k in upd_kinds or k in (2, 3, 10, 13, 14, 16):
            roof = lp["flops"][q] / (UPD3_STANDALONE_TFLOPS * 1e12) * 1e3
        else:
            roof = lp["bytes"][q] / 8e12 * 1e3
        bound += float(np.maximum(roof, fk).sum())
    by_kind = {KIND_NAMES.get(int(k), str(int(k))): {"launches": int((kinds == k).sum()), "ms": float(msl[kinds == k].sum()),
                                                      "floor_us": 1e3 * floor(k)} for k in sorted(set(kinds.tolist()))}
    return {"critical_path_ms": cp, "ms_per_step_over_critical_path": ms_step / cp if cp > 0 else None,
            "critical_path_detail": {"steps_of_64_columns_on_the_longest_path": path_steps, "step_floor_us": 1e3 * step_floor,
                                     "chain_ms": t_chain, "thin_levels_on_the_path": path_thin, "thin_launch_floor_us": 1e3 * f_thin,
                                     "thin_ms": t_thin, "update_flops": upd_flops, "update_ms_at_standalone_rate": t_upd,
                                     "standalone_update_TFLOPs": UPD3_STANDALONE_TFLOPS,
                                     "etree_levels": int(lv.max()) + 1 if nsuper else 0},
            "launch_list_bound_ms": bound, "ms_per_step_over_launch_list_bound": ms_step / bound if bound > 0 else None,
            "profiled_ms_by_kind": by_kind, "profiled_ms_total": float(msl.sum())}


def secondary_line(workload, m, steps=3, warmup=1):
    """One of the other single-GPU configurations of BASELINE.json (configs[1]: Poisson 100^3; SURVEY 8d's
    stand-ins of nd24k and G3_circuit), same step as the headline: resident refactorizations, then
    the profiled pass, the device solve, residual and factor invariants."""
    from suitesparse_amd import cholmod as ch
    from suitesparse_amd import generators as G
    n, Ap, Ai, Ax, stype, perm, wname = build_workload(workload, m)
    S = ch.Session(factor_on_device=True, ordering="default")
    A = S.sparse(n, Ap, Ai, Ax, stype)
    Lf = S.analyze(A, perm)
    fl = S.cm.fl
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    for _ in range(warmup):
        t0 = time.perf_counter()
        assert S.refactorize_resident(Lf) == 1
        dt0 = time.perf_counter() - t0
    # (a 5 ms step timed over three repetitions is noise: at least a quarter of a second per timing loop)
    steps = max(steps, min(100, int(0.25 / max(dt0, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(steps):
        assert S.refactorize_resident(Lf) == 1
    dt = (time.perf_counter() - t0) / steps
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    t0 = time.perf_counter()
    for _ in range(steps):
        assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    dt_api = (time.perf_counter() - t0) / steps
    stats = S.hip_stats(Lf)
    roof = roofline_of(S, Lf, wname, 1)
    b = G.demo_rhs(n)
    x = S.solve(Lf, b)
    r = G.sym_matvec(n, Ap, Ai, Ax, stype, x) - b
    checks = S.factor_checks(Lf)
    # quoted on SURVEY 8d's step: cholmod_l_factorize (A, L, Common) from the host matrix (CHOLMOD/Cholesky/
    # cholmod_factorize.c:97-288; the demo's convention, cholmod_l_demo.c:293, :691-692) -- values-only H2D + gather + the
    # factorization; the step on the resident S stands beside it
    out = {"workload": wname, "n": int(n), "fl": fl, "executed_flops": stats[1], "value": fl / dt_api / 1e9, "unit": "GFLOP/s",
           "ms_per_step": 1e3 * dt_api, "ms_per_step_api": 1e3 * dt_api, "ms_per_step_resident": 1e3 * dt, "steps": steps,
           "step": "cholmod_l_factorize(A, L, Common), A in host memory, L left in HBM (SURVEY 8d t_factorize)",
           "pct_fp64_mfma_peak": 100.0 * fl / dt_api / 1e12 / FP64_MFMA_PEAK_TFLOPS,
           "value_resident": fl / dt / 1e9, "pct_fp64_mfma_peak_resident": 100.0 * fl / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS,
           "launches_per_step": int(stats[2]), "Lx_GB": 8e-9 * ch.FactorView(Lf).xsize,
           "residual_2norm": float(np.linalg.norm(r) / np.linalg.norm(b)),
           "upper_nonzeros": checks["upper_nonzeros"], "nonfinite": checks["nonfinite"],
           "solve_device_ms": 1e3 * float(S.hip_stats(Lf)[24]), "roofline": roof}
    if roof is not None:
        try:
            out.update(critical_path(S, Lf, roofline_of.last_profile, 1e3 * dt))
        except Exception as e:          # (a diagnostic: never lose the line to it)
            out["critical_path_error"] = repr(e)
    if roof is not None:
        tf = roof["thin_front_kernel"]
        # the thin configuration is priced against HBM: algorithmic bytes of the whole factorization / time
        if workload == "poisson2d":
            algo = (roof["extend_add"]["algorithmic_GB"] + tf["algorithmic_GB"]) * 1e9 + 2.0 * stats[5]   # stats[5] = 8 B x entries of L
            tt = thin_traffic()
            if tt is not None:
                # the thin kernels against HBM: algorithmic bytes and time from this run, counter bytes from the committed passes
                tf["traffic_bytes_per_factorization"] = tt["traffic_bytes"]
                tf["traffic_over_algorithmic"] = tt["traffic_bytes"] / max(tf["algorithmic_GB"] * 1e9, 1.0)
                tf["traffic_detail"] = tt
            out["hbm_roofline"] = {"bound": "hbm", "achieved": algo / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": algo / dt / 8e12, "traffic": tt["traffic_bytes"] if tt else None,
                                   "traffic_note": "HBM-side bytes of the thin-front kernels only (FETCH_SIZE x 2 + WRITE_SIZE over one refactorization); "
                                                   "their algorithmic bytes: roofline.thin_front_kernel.algorithmic_GB" if tt else None,
                                   "algorithmic_bytes": "thin fronts (children in, panel + block out) + extend-add of the generic "
                                                        "fronts + 16 B per entry of L for the generic panels"}
    if workload in ("poisson3d", "poisson2d"):
        ld = G.poisson_logdet(*([m] * (3 if workload == "poisson3d" else 2)))
        out["logdet_rel_err"] = abs(2.0 * checks["half_logdet"] - ld) / abs(ld)
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()
    return out


def complex_line(m=64, steps=3):
    """Complex (Hermitian positive definite) input through the same calls: Poisson m^3's pattern with random
    phases on the off-diagonals.  The engine computes on the index space of the real twin [re -im; im re]
     and keeps the factor in complex storage (the even twin columns only = the interleaved complex
    factor, 2 xsize doubles; CHOLMOD_HIP_CX_STORAGE); its update kernels contract over the even panel columns
    (a complex multiply-add as four real ones).  Beside it the full twin with the same update kernels
    (CHOLMOD_HIP_CX_TWIN=1, 4 xsize doubles), the plain embedding (eight real multiply-adds,
    CHOLMOD_HIP_TWIN_FULL_K=1) and the real factorization of the same pattern."""
    from suitesparse_amd import cholmod as ch
    from suitesparse_amd import generators as G
    n, Ap, Ai, Ax = G.poisson3d(m)
    perm = G.geometric_nd(m, m, m, 4)
    Az = G.hermitian_phases(n, Ap, Ai, Ax, seed=m)
    out = {"workload": f"hermitian_poisson3d_{m}^3_geometricND_leaf4 (complex input)", "n": int(n)}
    for tag, vals in (("real", Ax), ("complex", Az), ("complex_full_twin", Az), ("complex_plain_embedding", Az)):
        if tag == "complex_plain_embedding":
            os.environ["CHOLMOD_HIP_TWIN_FULL_K"] = "1"
        if tag == "complex_full_twin":
            os.environ["CHOLMOD_HIP_CX_TWIN"] = "1"
        try:
            S = ch.Session(factor_on_device=True, ordering="default")
            A = S.sparse(n, Ap, Ai, vals, -1)
            Lf = S.analyze(A, perm)
            fl = S.cm.fl
            assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
            assert S.refactorize_resident(Lf) == 1
            t0 = time.perf_counter()
            for _ in range(steps):
                assert S.refactorize_resident(Lf) == 1
            dt = (time.perf_counter() - t0) / steps
            cx = tag != "real"
            b = G.demo_rhs(n).astype(np.complex128 if cx else np.float64)
            x = S.solve(Lf, b)
            r = (G.herm_matvec(n, Ap, Ai, vals, x) if cx else G.sym_matvec(n, Ap, Ai, vals, -1, x)) - b
            out[tag] = {"ms_per_step": 1e3 * dt, "fl_real_convention": fl,
                        "residual_2norm": float(np.linalg.norm(r) / np.linalg.norm(b))}
            if cx:
                # a complex multiply-add = 4 real ones = 8 flop: the complex factorization is 4 fl
                out[tag]["TFLOPs_on_4fl"] = 4.0 * fl / dt / 1e12
                T = C.cast(Lf.contents.cx_twin, C.POINTER(ch.Factor))
                st = np.zeros(ch.CHOLMOD_HIP_NSTATS)
                S.L.cholmod_hip_get_stats(T.contents.hip_plan, st.ctypes.data)
                out[tag]["factor_GB_in_HBM"] = st[5] / 1e9
                out[tag]["arena_GB"] = st[4] / 1e9
            S.free_factor(Lf)
            S.free_sparse(A)
            S.finish()
        finally:
            os.environ.pop("CHOLMOD_HIP_TWIN_FULL_K", None)
            os.environ.pop("CHOLMOD_HIP_CX_TWIN", None)
    out["complex_over_real_time"] = out["complex"]["ms_per_step"] / out["real"]["ms_per_step"]
    out["complex_storage_over_plain_embedding_time"] = out["complex"]["ms_per_step"] / out["complex_plain_embedding"]["ms_per_step"]
    out["full_twin_over_plain_embedding_time"] = out["complex_full_twin"]["ms_per_step"] / out["complex_plain_embedding"]["ms_per_step"]
    out["note"] = ("a complex multiply-add is 4 real ones: zherk / zgemm cost 4x the real update flops of the same pattern (what the "
                   "even-column update kernels execute), the plain embedding 8x; complex storage = the reference's interleaved L->x in "
                   "HBM (half the twin's bytes in L, contribution blocks and extend-add); the panel chain runs on the twin's index space")
    return out


def visible_devices():
    """HIP devices this process can see (the engine's own probe: no torch, no device context)."""
    from suitesparse_amd import cholmod as ch
    cnt = C.c_int(0)
    return int(cnt.value) if ch.lib().cholmod_hip_device_count(C.byref(cnt)) == 0 else 0


def launch_ranks(n, backend):
    """`python bench.py --gpus N ...` with no launcher around it: re-run this command line as N ranks
    under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and hand its exit code
    back.  The ranks' stdout is this process's stdout, so the one JSON line of rank 0 is the output."""
    import socket
    import subprocess
    if backend != "gloo":
        have = visible_devices()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible HIP devices, this node shows {have} "
                  "(--dist-backend gloo lets the ranks share device 0, for tests)", file=sys.stderr)
            return 2
    # (a port below the ephemeral range: one the kernel hands out for "bind to 0" can be taken by any outgoing connection
    # of the host between this probe and the rendezvous)
    import random
    port = None
    for _ in range(200):
        cand = random.randrange(20000, 32000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", cand))
            port = cand
        except OSError:
            pass
        finally:
            s.close()
        if port is not None:
            break
    if port is None:
        print("bench.py: no free rendezvous port on 127.0.0.1", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    return subprocess.call(cmd, env=env)


LINE_LIMIT = 4096       # bytes of the ONE stdout line (round-5 review: a 25.9 KB line left the driver's record unparsed)


def _r(x, sig=6):
    """A number at `sig` significant digits (None / non-finite -> None): the line carries numbers, not 17-digit reprs."""
    if x is None or isinstance(x, (bool, str)):
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{sig}g}")


def compact_line(full):
    """The ONE line the driver parses, built from the full record: numbers only, no prose, <= LINE_LIMIT bytes.  The full
    record (counter tables, per-class times, sweeps, notes) goes to bench_detail.json and to stderr."""
    g = full.get
    line = {k: g(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                              "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(g("value"), 12), _r(g("ms_per_step"), 12)
    cfg = g("config") or {}
    line["config"] = {k: _r(cfg.get(k), 12 if k == "fl" else 7) for k in ("workload", "n", "fl", "executed_flops", "nsuper", "Lx_GB", "launches_per_step")
                      if k in cfg}
    line["config"]["parallelism"] = ("1 GPU" if g("n_gpus") == 1 else f"{g('n_gpus')} GPUs, etree subtrees + column slabs of shared fronts")
    for k in ("pct_fp64_mfma_peak_per_gpu", "ms_per_step_resident", "ms_per_step_api", "value_api", "residual_2norm",
              "measured_fp64_mfma_ceiling_TFLOPs"):
        if g(k) is not None:
            line[k] = _r(g(k))
    rf = g("roofline")
    if rf:
        mu = rf.get("mfma_utilisation")
        if isinstance(mu, dict):
            mu = mu.get("mfma_pipe_utilisation", mu.get("mfma_utilisation"))
        ach = _r(rf["achieved"], 8)
        line["roofline"] = {"bound": rf["bound"], "achieved": ach, "peak": rf["peak"], "unit": rf["unit"],
                            "frac": (ach / rf["peak"]) if ach is not None else None, "traffic": _r(rf.get("traffic")),
                            "traffic_over_algorithmic": _r(rf.get("traffic_over_algorithmic"), 4),
                            "mfma_utilisation": _r(mu, 4), "kernel": (rf.get("kernel") or "").split(" ")[0],
                            "launches": rf.get("launches"), "avg_launch_ms": _r(rf.get("avg_launch_ms"))}
    else:
        line["roofline"] = None
    cb = g("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                "host_cores": cb.get("host_cores"), "cpu_quota": cb.get("cpu_quota"), "kind": cb.get("kind"),
                                "sample": (cb.get("sample_short") or (cb.get("sample") or "")[:60])}
        if cb.get("by_threads"):
            line["cpu_baseline"]["by_threads"] = {str(q["threads"]): _r(q["GFLOPs"], 4) for q in cb["by_threads"]}
    else:
        line["cpu_baseline"] = None
    hs = g("host_seconds")
    if hs:
        line["host_seconds"] = {k: _r(v, 4) for k, v in hs.items()}
    for key in ("factor_checks", "factor_checks_distributed"):
        fc = g(key)
        if fc:
            line[key] = {k: _r(fc.get(k), 4) for k in ("logdet_rel_err", "trace_rel_err", "upper_nonzeros", "nonfinite",
                                                       "nonpositive_diag", "solve_device_ms", "ranks_that_could_not_check")
                         if fc.get(k) is not None}
    sec = []
    for s in g("secondary") or []:
        if "error" in s:
            sec.append({"workload": s.get("workload", "")[:40], "error": str(s["error"])[:80]})
            continue
        if "complex" in s:          # the complex line: times only
            sec.append({"workload": s["workload"].split(" ")[0], "ms_per_step": _r(s["complex"]["ms_per_step"]),
                        "TFLOPs_on_4fl": _r(s["complex"].get("TFLOPs_on_4fl")), "complex_over_real_time": _r(s.get("complex_over_real_time"), 4),
                        "residual_2norm": _r(s["complex"].get("residual_2norm"), 3)})
            continue
        r2 = s.get("roofline") or {}
        o = {"workload": s["workload"].split(" ")[0], "value": _r(s.get("value")), "ms_per_step": _r(s.get("ms_per_step")),
             "ms_per_step_resident": _r(s.get("ms_per_step_resident")), "pct_fp64_mfma_peak": _r(s.get("pct_fp64_mfma_peak"), 4),
             "pct_fp64_mfma_peak_resident": _r(s.get("pct_fp64_mfma_peak_resident"), 4),
             "roofline_frac": _r(r2.get("frac"), 4), "critical_path_ratio": _r(s.get("ms_per_step_over_critical_path"), 4),
             "residual_2norm": _r(s.get("residual_2norm"), 3)}
        if s.get("hbm_roofline"):
            o["hbm_roofline_frac"] = _r(s["hbm_roofline"].get("frac"), 4)
        sec.append(o)
    if sec:
        line["secondary"] = sec
    ex = g("exchange")
    if ex:
        line["exchange"] = {k: _r(v, 5) for k, v in ex.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
    for k in ("truncated", "error", "residual_2norm_note"):
        if g(k):
            line[k] = str(g(k))[:160]
    if g("error"):      # the watchdog's line: where the run stood (small: nothing after the timed region exists)
        for k in ("phase", "rank", "steps_completed", "last_completed_step_ms", "seconds_since_start"):
            if k in full:
                line[k] = _r(g(k)) if not isinstance(g(k), str) else g(k)[:160]
        if g("completed_step_ms"):
            line["completed_step_ms"] = [_r(v, 5) for v in g("completed_step_ms")[-8:]]
        if g("progress"):
            line["progress"] = g("progress")
        for k in ("roofline", "cpu_baseline"):
            if line.get(k) is None:
                line.pop(k, None)
    line["detail"] = "bench_detail.json"
    txt = json.dumps(line, allow_nan=False, separators=(",", ":"))
    # (should anything above ever grow past the limit, the optional parts go first: the contract keys never do)
    for drop in ("secondary", "host_seconds", "factor_checks", "factor_checks_distributed", "exchange"):
        if len(txt) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        txt = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(txt) <= LINE_LIMIT, len(txt)
    return txt


def emit(full, out_fd):
    """Full record -> bench_detail.json (beside bench.py, and under gpurun_out/ when that exists) and stderr; the compact
    line -> stdout."""
    def clean(o):
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, (float, np.floating)):
            o = float(o)
            return o if (o == o and abs(o) != float("inf")) else None
        return o
    full = clean(full)
    detail = json.dumps(full, allow_nan=False, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
            except OSError:
                pass
    sys.stderr.write("[bench detail] " + json.dumps(full, allow_nan=False) + "\n")
    sys.stderr.flush()
    os.write(out_fd, (compact_line(full) + "\n").encode())


METRIC = "GFLOP/s supernodal Cholesky factor (Common->fl / t_factorize)"


class Watchdog:
    """A hung collective (or a rank that died) must end in a JSON line with "error", not in the driver's kill: a thread per
    rank watches a per-phase deadline and an overall one.  On expiry -- or when the launcher sends SIGTERM because a peer
    failed -- rank 0 prints the error line (phase, steps completed, time of the last completed step, and which exchange of
    the factorization its device has entered and not left: cholmod_hip_progress) and every rank exits non-zero; the other
    ranks report their own progress on stderr and give rank 0 a few seconds' head start so that its line gets out before
    the launcher tears the job down."""

    def __init__(self, rank, world, out_fd, overall_s):
        import threading
        self.rank, self.world, self.out_fd = rank, world, out_fd
        self.t_start = time.monotonic()
        self.overall = overall_s
        self.phase, self.deadline = "start", None
        self.steps_done, self.last_step_ms, self.step_ms = 0, None, []
        self.lib, self.plan, self.info, self.fallback = None, None, {}, None
        self.done = False
        self._lock = threading.Lock()
        self._rfd = None
        try:        # SIGTERM / SIGINT while the main thread sits in a native call: a wakeup pipe this thread reads
            import signal
            r, w = os.pipe()
            os.set_blocking(w, False)
            os.set_blocking(r, False)
            signal.signal(signal.SIGTERM, lambda *a: None)
            signal.set_wakeup_fd(w, warn_on_full_buffer=False)
            self._rfd = r
        except Exception:
            pass
        self._thr = threading.Thread(target=self._run, daemon=True)
        self._thr.start()

    def arm(self, phase, seconds):
        self.phase = phase
        self.deadline = None if seconds is None else time.monotonic() + seconds

    def step_done(self, ms):
        self.steps_done += 1
        self.last_step_ms = ms
        self.step_ms.append(ms)

    def finish(self):
        self.done = True

    def progress(self):
        if self.lib is None or not self.plan:
            return None
        o = (C.c_int64 * 12)()
        if self.lib.cholmod_hip_progress(self.plan, o) != 0:
            return None
        d = {"factorizations_started": o[0], "launches_enqueued": o[1], "launches_in_schedule": o[2],
             "exchanges_enqueued": o[3], "exchanges_in_schedule": o[4],
             "exchange_entered_on_device": o[5], "exchange_left_on_device": o[6]}
        if o[5] > o[6]:
            d["pending_exchange"] = {"sequence": o[5], "collective": {7: "reduce-scatter (+ broadcast of the diagonal block)",
                                                                      11: "all-gather"}.get(o[7], str(o[7])),
                                     "rank_group": [o[8], o[8] + o[9] - 1], "block_column_width": o[10], "rows_below": o[11]}
        return d

    def _fire(self, why):
        with self._lock:
            if self.done:
                return
            self.done = True
        line = {"metric": METRIC, "value": None, "unit": "GFLOP/s", "n_gpus": self.world, "higher_is_better": True,
                "error": why, "phase": self.phase, "rank": self.rank,
                "seconds_since_start": time.monotonic() - self.t_start,
                "steps_completed": self.steps_done, "last_completed_step_ms": self.last_step_ms,
                "completed_step_ms": self.step_ms, "progress": self.progress()}
        line.update(self.info)
        rc = 3
        if self.fallback is not None:
            # the timed region is complete: the measurement goes out, marked as cut short
            line = dict(self.fallback, truncated=why + " -- sections after the timed region are missing from this line")
            rc = 0
        txt = json.dumps(line, default=str)
        if self.rank == 0:
            try:
                os.write(self.out_fd, (compact_line(line) + "\n").encode())
            except Exception:
                os.write(self.out_fd, (json.dumps({k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "error", "phase")},
                                                  default=str)[:LINE_LIMIT] + "\n").encode())
        else:
            time.sleep(8.0)             # rank 0's line first
        sys.stderr.write(f"[bench watchdog rank {self.rank}] {txt}\n")
        sys.stderr.flush()
        os._exit(rc)

    def _run(self):
        import select
        while not self.done:
            if self._rfd is not None:
                r, _, _ = select.select([self._rfd], [], [], 0.5)
                if r:
                    try:
                        sig = os.read(self._rfd, 64)
                    except OSError:
                        sig = b""
                    if sig and not self.done:
                        self._fire("signal %s from the launcher (a peer rank failed or the job was cancelled) during phase '%s'"
                                   % (",".join(str(b) for b in sig), self.phase))
            else:
                time.sleep(0.5)
            now = time.monotonic()
            if self.done:
                return
            if self.deadline is not None and now > self.deadline:
                self._fire(f"deadline of phase '{self.phase}' exceeded")
            if self.overall and now - self.t_start > self.overall:
                self._fire(f"overall deadline of {self.overall:.0f} s exceeded in phase '{self.phase}'")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="poisson3d")
    ap.add_argument("--grid", "--m", dest="m", type=int, default=0,
                    help="grid points per side (default: 200, falling back to 160 / 100 if HBM is short)")
    ap.add_argument("--cpu-sample-m", type=int, default=100,
                    help="CPU baseline: Poisson m^3 through the product's CPU path (default 100 = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other single-GPU configurations (Poisson 100^3, the nd24k and G3_circuit stand-ins) "
                         "that follow the headline workload in the same JSON line")
    ap.add_argument("--check", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the solve / residual / factor invariants after the timed region")
    ap.add_argument("--hip-flags", type=int, default=0)
    ap.add_argument("--matrix", default=None,
                    help="factor a symmetric positive definite Matrix Market / triplet file instead of a synthetic "
                         "workload (e.g. $SSGET/ND/nd24k.mtx when present), ordered by the built-in nested dissection")
    ap.add_argument("--ordering", default="geometric", choices=["geometric", "builtin"],
                    help="geometric = the nested dissection SURVEY 8d prescribes for the metric (default); "
                         "builtin = cholmod_l_analyze's own ordering (host/order.c), for information")
    ap.add_argument("--deadline", type=float, default=1500.0,
                    help="seconds after which a run that has not printed its line prints an error line instead and exits 3 "
                         "(0 = never); phases (attach, first factorization, every timed step) have deadlines of their own")
    ap.add_argument("--step-deadline", type=float, default=0.0,
                    help="seconds one factorization step may take (default: max (120, 10 x the first factorization))")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL) or gloo (ranks sharing a GPU, tests)")
    ap.add_argument("--exchange", default="native", choices=["native", "callback"],
                    help="native: the engine calls RCCL itself (default with the nccl backend); callback: "
                         "torch.distributed all_reduce through the host callback")
    args = ap.parse_args()

    # --gpus N without an outer launcher (WORLD_SIZE unset): this process becomes the launcher and
    # starts N ranks of itself under torch.distributed.run, one per GPU, exactly as the driver's
    # N > 1 command line does; rank 0 of those prints the line.  Under a launcher --gpus must agree
    # with WORLD_SIZE, and a run that cannot give every rank a device of its own fails loudly.
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, args.dist_backend))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} under a launcher with WORLD_SIZE={os.environ['WORLD_SIZE']}: "
                         "one rank per GPU, the two must agree")

    # stdout carries exactly ONE JSON line: native libraries that print there (RCCL's
    # version banner at communicator creation) are pointed at stderr for the whole run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    wd = Watchdog(rank, world, real_stdout, args.deadline)
    wd.arm("process group / device set-up", 600)
    dist = None
    # CHOLMOD_HIP_SHARE_AS_WORLD=k with one rank: the engine's self test of the
    # exchange path (fronts a k-rank run would share go through pack / RCCL
    # all-reduce / unpack); measures that path's overhead on a 1-GPU box
    selftest = world == 1 and int(os.environ.get("CHOLMOD_HIP_SHARE_AS_WORLD", "0")) > 1
    if selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 and args.dist_backend != "gloo" and visible_devices() < world:
        raise SystemExit(f"bench.py: {world} ranks need {world} visible HIP devices, this node shows {visible_devices()}")
    if world > 1 or selftest:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "gloo":
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from suitesparse_amd import cholmod as ch
    lib = ch.lib()
    if lib.cholmod_hip_probe() != 1:
        raise RuntimeError("bench.py needs a HIP device; there is no CPU path to measure")
    if lib.cholmod_hip_set_device(local_rank) != 0:
        raise RuntimeError(f"bench.py: rank {rank} cannot select HIP device {local_rank}")

    # exchange: the engine's native RCCL path (stream-ordered ncclAllReduce on its own
    # streams) with the nccl backend; the torch.distributed callback with gloo (ranks
    # sharing a GPU in tests), or on request (--exchange callback)
    allreduce = None
    native = dist is not None and args.dist_backend == "nccl" and args.exchange == "native"
    if dist is not None and not native:
        from suitesparse_amd.dist import make_allreduce
        allreduce = make_allreduce(subgroups=None)     # the plan's own rank groups are created below
    grids = [0] if args.matrix else [args.m] if args.m > 0 else ([200, 160, 100] if args.workload == "poisson3d" else [100])
    S = None
    for gi, m in enumerate(grids):
        t0 = time.perf_counter()
        if args.matrix:
            from suitesparse_amd import generators as G
            n, Ap, Ai, Ax, stype = G.read_triplet(args.matrix)
            if stype == 0:
                raise SystemExit("bench.py --matrix needs a symmetric file (one triangle stored)")
            perm, wname = None, os.path.basename(args.matrix) + "_builtinND"
        else:
            n, Ap, Ai, Ax, stype, perm, wname = build_workload(args.workload, m)
        t_gen = time.perf_counter() - t0
        S = ch.Session(factor_on_device=True, hip_flags=args.hip_flags, rank=rank, world=world,
                       allreduce=allreduce, ordering="default")
        if args.ordering == "builtin":
            perm = None
            wname = wname.split("_geometricND")[0] + "_builtinND"

        wd.info = {"config": {"workload": wname, "n": int(n)}}
        wd.arm(f"analyze ({wname})", 900)
        A = S.sparse(n, Ap, Ai, Ax, stype)
        t0 = time.perf_counter()
        Lf = S.analyze(A, perm)
        t_analyze = time.perf_counter() - t0
        # (one GPU: the engine's plan -- schedule, maps, the HBM reservation -- is built inside cholmod_l_analyze, as the
        # reference cuts its device pools there; several ranks: by cholmod_l_hip_prepare below)
        plan_in_analyze = bool(ch.FactorView(Lf).hip_plan)
        t_plan = float(S.cm.hip_plan_seconds) if plan_in_analyze else 0.0
        fl = S.cm.fl
        fv = ch.FactorView(Lf)
        # reserve HBM for L and the contribution blocks (no collective in here)
        t0 = time.perf_counter()
        wd.arm("plan build + reservation of HBM (cholmod_l_hip_prepare)", 600)
        ok = S.L.cholmod_l_hip_prepare(Lf, C.byref(S.cm))
        short = (ok != 1 and S.cm.status == ch.OUT_OF_MEMORY)
        if dist is not None:
            # every rank must take the same decision
            import torch
            flag = torch.tensor([1.0 if short else 0.0], device="cuda")
            dist.all_reduce(flag)
            short = flag.item() > 0
        if short and gi + 1 < len(grids):
            if rank == 0:
                print(f"[bench] {wname}: not enough HBM, falling back to grid {grids[gi + 1]}", file=sys.stderr)
            S.free_factor(Lf)
            S.free_sparse(A)
            S.finish()
            continue
        assert ok == 1 and S.cm.status == ch.OK, (ok, S.cm.status)
        wd.lib, wd.plan = S.L, ch.FactorView(Lf).hip_plan
        if world > 1 or selftest:
            S.L.cholmod_hip_progress_enable(wd.plan, 1)     # (markers around every exchange: what a hung rank was waiting in)
        if native:
            wd.arm("RCCL attach (ncclCommInitRank + one ncclCommSplit per rank group)", 420)
            # one 128-byte RCCL id from rank 0 to everybody, then every rank attaches its plan
            import torch
            idb = np.zeros(128, dtype=np.uint8)
            if rank == 0:
                assert S.L.cholmod_hip_rccl_unique_id(idb.ctypes.data) == 0
            t = torch.from_numpy(idb).cuda()
            dist.broadcast(t, 0)
            idb = t.cpu().numpy().copy()
            rc = S.L.cholmod_hip_rccl_attach(ch.FactorView(Lf).hip_plan, idb.ctypes.data)
            # every rank must agree that the native path is up; otherwise all of them take the callback
            flag = torch.tensor([0.0 if rc == 0 else 1.0], device="cuda")
            dist.all_reduce(flag)
            if flag.item() > 0:
                if rank == 0:
                    print(f"[bench] cholmod_hip_rccl_attach failed on some rank (this rank: {rc}): "
                          "falling back to the torch.distributed callback", file=sys.stderr)
                S.L.cholmod_hip_rccl_detach(ch.FactorView(Lf).hip_plan)
                from suitesparse_amd.dist import make_allreduce
                native = False
                allreduce = make_allreduce(subgroups=None)
                S._keep.append(allreduce)
                assert S.L.cholmod_hip_set_allreduce(ch.FactorView(Lf).hip_plan, allreduce, None) == 0
        if allreduce is not None and world > 1:
            # process groups for the rank ranges this plan shares fronts over
            # (same partition on every rank -> same collective new_group calls)
            g0 = np.empty(fv.nsuper, dtype=np.int64)
            gn = np.empty(fv.nsuper, dtype=np.int64)
            assert S.L.cholmod_hip_get_groups(ch.FactorView(Lf).hip_plan, g0.ctypes.data, gn.ctypes.data) == 0
            allreduce.create_groups(sorted(set(zip(g0[gn > 1].tolist(), gn[gn > 1].tolist()))))
        # first factorization: uploads S (H2D, outside the timed region)
        wd.arm("first factorization (upload of S, first pass over every collective)", 600)
        tf0 = time.perf_counter()
        ok = S.factorize(A, Lf)
        t_first_fact = time.perf_counter() - tf0
        t_first = time.perf_counter() - t0
        assert ok == 1 and S.cm.status == ch.OK, (ok, S.cm.status)
        break

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    step_dl = args.step_deadline if args.step_deadline > 0 else max(120.0, 10.0 * t_first_fact)
    for k in range(max(args.warmup - 1, 0)):
        wd.arm(f"warm-up step {k + 1}", step_dl)
        assert S.refactorize_resident(Lf) == 1
    wd.arm("barrier before the timed region", step_dl)
    barrier()
    t0 = time.perf_counter()
    dev_s = 0.0
    for k in range(args.steps):
        wd.arm(f"timed step {k + 1} of {args.steps}", step_dl)
        ts = time.perf_counter()
        assert S.refactorize_resident(Lf) == 1     # synchronises the engine stream
        wd.step_done(1e3 * (time.perf_counter() - ts))
        dev_s += S.hip_stats(Lf)[0]
    wd.arm("barrier after the timed region", step_dl)
    barrier()
    elapsed = time.perf_counter() - t0
    # from here on the measurement exists: should the sections that follow (API steps, profiled pass, checks, CPU baseline,
    # secondary workloads) run into the overall deadline, the watchdog prints THIS line (marked truncated) instead of an error
    wd.fallback = {"metric": METRIC, "value": fl * args.steps / elapsed / 1e9, "unit": "GFLOP/s", "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                   "data": "file" if args.matrix else "synthetic", "config": {"workload": wname, "n": int(n), "fl": fl},
                   "ms_per_step_resident": 1e3 * elapsed / args.steps}
    wd.arm("after the timed region (API steps, profiled pass, checks, secondary workloads)", None)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = S.hip_stats(Lf)
    exec_flops = stats[1]

    # the same step through the public API (SURVEY 8d's t_factorize): cholmod_l_factorize
    # from the host matrix.  The first call permutes A into S = tril(PAP') on the host cores
    # and uploads S; calls with the same pattern (these) send A->x only and gather it into
    # the resident S on the device; L stays in HBM
    # (one untimed call first: the pinned staging buffer of the value upload is allocated by the first call of this kind;
    # a short step is repeated until the loop lasts a quarter of a second)
    assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    api_steps = max(min(args.steps, 3), min(50, int(0.25 * args.steps / max(elapsed, 1e-6))))
    barrier()
    t0 = time.perf_counter()
    for _ in range(api_steps):
        assert S.factorize(A, Lf) == 1 and S.cm.status == ch.OK
    barrier()
    elapsed_api = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed_api], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_api = float(t.item())

    roof = None
    if not args.no_profile_pass:
        roof = roofline_of(S, Lf, wname, world)

    # correctness of what was just timed (outside the timed region): device solve ->
    # residual, and one pass over the resident factor for its invariants; for the
    # Poisson grids log det(A) is known in closed form (a checksum of the whole factor)
    resid, checks = None, None
    check_note = None
    do_check = not args.no_check
    dist_checks = None
    if not args.no_check and world > 1:
        # invariants of the factor as it lies distributed over the ranks (every front checked by the first rank
        # of its group, the five sums all-reduced): no gathered copy, so this works where the gather does not fit
        import torch
        from suitesparse_amd import generators as G
        try:
            mine = np.concatenate([S.factor_checks_local(Lf), [0.0]])
        except Exception:               # (a rank that cannot check still takes part in the sum, and says so)
            mine = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
        loc = torch.from_numpy(mine).cuda()
        dist.all_reduce(loc)
        v = loc.cpu().numpy()
        dist_checks = dict(half_logdet=float(v[0]), upper_nonzeros=int(v[1]), nonfinite=int(v[2]), fro2=float(v[3]), nonpositive_diag=int(v[4]),
                           ranks_that_could_not_check=int(v[5]))
        if not args.matrix and args.workload in ("poisson3d", "poisson2d"):
            ld = G.poisson_logdet(*([m] * (3 if args.workload == "poisson3d" else 2)))
            dist_checks["logdet_closed_form"] = ld
            dist_checks["logdet_rel_err"] = abs(2.0 * dist_checks["half_logdet"] - ld) / abs(ld)
            # ||L||_F^2 = trace (A) = 2 d n for the d-dimensional Dirichlet Laplacian
            tr = 2.0 * (3 if args.workload == "poisson3d" else 2) * n
            dist_checks["trace_rel_err"] = abs(dist_checks["fro2"] - tr) / tr
    if do_check and world > 1:
        # the complete factor on every rank, next to the rank's own part: 181.6 GB + 117 GB at two ranks leave 8 GB of
        # one MI355X; the engine makes room (arena, then the rank's part through host memory) and its ranks agree on
        # the outcome -- if there is no room, the line is printed without the solve / factor checks rather than lost
        import torch
        eh = S.cm.error_handler
        S.cm.error_handler = ch.ERRFUNC(0)
        got = S.L.cholmod_l_gather_factor(Lf, C.byref(S.cm)) == 1
        S.cm.error_handler = eh
        flag = torch.tensor([0.0 if got else 1.0], device="cuda")
        dist.all_reduce(flag)
        if flag.item() > 0:
            do_check = False
            check_note = ("skipped: the gathered factor does not fit next to the rank's own part of L on every rank "
                          f"(status {S.cm.status} on rank {rank})")
            S.cm.status = ch.OK
    if do_check:
        from suitesparse_amd import generators as G
        t0 = time.perf_counter()
        b = G.demo_rhs(n)
        x = S.solve(Lf, b)
        t_solve = time.perf_counter() - t0
        r = G.sym_matvec(n, Ap, Ai, Ax, stype, x) - b
        resid = float(np.linalg.norm(r) / np.linalg.norm(b))
        checks = S.factor_checks(Lf)
        checks["solve_seconds_incl_h2d_d2h"] = t_solve
        checks["solve_device_ms"] = 1e3 * float(S.hip_stats(Lf)[24])      # L and L' sweeps, kernels only
        if not args.matrix and args.workload in ("poisson3d", "poisson2d"):
            ld = G.poisson_logdet(*([m] * (3 if args.workload == "poisson3d" else 2)))
            checks["logdet_closed_form"] = ld
            checks["logdet_rel_err"] = abs(2.0 * checks["half_logdet"] - ld) / abs(ld)

    # everything the line needs from the headline factor, then release its HBM (181.6 GB + arena at
    # 200^3): the secondary workloads and the micro-benchmarks below need the room
    nsuper, xsize = int(fv.nsuper), int(fv.xsize)
    S.free_factor(Lf)
    S.free_sparse(A)
    S.finish()
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_sample_m)
        # the other single-GPU configurations of BASELINE.json, in the same record (a few seconds each)
        secondary = []
        if world == 1 and not args.no_secondary and not args.matrix and args.workload == "poisson3d" and m >= 160:
            # (box3d 50: the nd24k stand-in's stencil on a grid whose factorization has the flop count SURVEY 8d records for the
            # 42^3 stand-in under AMD, 2.4e12 -- under this build's nested dissection 42^3 itself is 8.0e11, a third of it)
            for wl, mm in (("poisson3d", 100), ("box3d", 42), ("poisson2d", 1259), ("box3d", 50)):
                try:
                    secondary.append(secondary_line(wl, mm))
                except Exception as e:          # never lose the headline line to a secondary workload
                    secondary.append({"workload": f"{wl} {mm}", "error": repr(e)})
            try:
                secondary.append(complex_line(64))
            except Exception as e:
                secondary.append({"workload": "hermitian_poisson3d_64", "error": repr(e)})
        pr = ch.probes()            # micro-benchmarks: lib/libcholmod_amd_probes.so, not the product library
        mf = pr.cholmod_hip_bench_update_kernel(8192, 8192, 512, 3, 8192)
        mf_big = pr.cholmod_hip_bench_update_kernel(16384, 16384, 4096, 2, 8192)
        mf2_big = pr.cholmod_hip_bench_update_kernel(16384, 16384, 4096, 2, 0)
        # The fp64 matrix-core ceiling, measured: an inline-assembly v_mfma_f64_16x16x4 loop with
        # nothing else in it, ~0.25 s per point so that the clock settles (waves per SIMD x
        # accumulators x operand data: full-mantissa operands cost power, i.e. clock).  Issue
        # rate = 64 cycles per MFMA per SIMD exactly, so the ceiling is 78.6 TFLOP/s x clock / 2.4 GHz.
        import ctypes as C2
        sweep = {}
        for zero in (0, 1):
            for nacc in (4, 8):
                for w in (1, 2):
                    o3 = (C2.c_double * 3)()
                    r = pr.cholmod_hip_bench_mfma_ceiling(w, nacc, int(0.25 * 2.3e9 / (4 * nacc * 64 * w)), zero, o3)
                    if r > 0:
                        sweep[f"{'zero' if zero else 'data'}_acc{nacc}_waves{w}"] = {
                            "TFLOPs": r / 1e12, "clock_GHz": o3[1], "cycles_per_mfma_per_simd": o3[0] if w == 1 else None}
        data_pts = [v["TFLOPs"] for k, v in sweep.items() if k.startswith("data")]
        mpeak = max(data_pts) * 1e12 if data_pts else 0.0
        value = fl * args.steps / elapsed / 1e9        # one job, all ranks together
        line = {
            "metric": METRIC,
            "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "file" if args.matrix else "synthetic",
            "config": {"workload": wname, "n": int(n), "nnz_lower": int(Ap[-1]),
                       "fl": fl, "executed_flops": exec_flops, "nsuper": nsuper,
                       "Lx_GB": 8e-9 * xsize, "arena_GB": 1e-9 * stats[4],
                       "levels": int(stats[3]), "launches_per_step": int(stats[2]),
                       "parallelism": "1 GPU" if world == 1 else
                       f"{world} GPUs: etree subtrees per rank + shared top fronts distributed by column slabs, "
                       f"{int(stats[17])} block-column exchanges (reduce-scatter / all-gather) per factorization "
                       f"({'RCCL, engine-native' if native else args.dist_backend + ' callback'})",
                       "input": "S=tril(PAP') resident in HBM; factor left in HBM"},
            "pct_fp64_mfma_peak_per_gpu": 100.0 * value / world / (1e3 * FP64_MFMA_PEAK_TFLOPS),
            "device_ms_per_step": 1e3 * dev_s / args.steps,
            "measured_update_kernel_TFLOPs_8192x8192x512": mf / 1e12 if mf > 0 else None,
            "measured_update_kernel_TFLOPs_16384x16384x4096": mf_big / 1e12 if mf_big > 0 else None,
            "measured_update_kernel_note": "k_update3 (one wave per 64x64 tile), the kernel of the big update regions; "
                                           "k_update2 (four waves per tile, LDS-staged) on the large shape: %s TFLOP/s"
                                           % (("%.1f" % (mf2_big / 1e12)) if mf2_big > 0 else "n/a"),
            "measured_fp64_mfma_ceiling_TFLOPs": mpeak / 1e12 if mpeak > 0 else None,
            "measured_fp64_mfma_ceiling_note": "best full-mantissa point of the inline-assembly MFMA loop (tools/mfma_ceiling.py): "
                                               "the issue rate is 64 cycles per MFMA per SIMD = the 78.6 TFLOP/s spec at 2.4 GHz, "
                                               "the figure is that rate at the clock the part sustains under the load",
            "mfma_ceiling_sweep": sweep,
            "ms_per_step_resident": 1e3 * elapsed / args.steps,
            "ms_per_step_api": 1e3 * elapsed_api / api_steps,
            "value_api": fl * api_steps / elapsed_api / 1e9,
            "pct_fp64_mfma_peak_api_per_gpu": 100.0 * fl * api_steps / elapsed_api / 1e12 / world / FP64_MFMA_PEAK_TFLOPS,
            "value_note": "`value` / `ms_per_step` time the step with S = tril(PAP') already resident in HBM (the bench contract: inputs "
                          "resident when the timed region starts); `value_api` / `ms_per_step_api` time cholmod_l_factorize (A, L, Common) "
                          "with A in host memory (SURVEY 8d's t_factorize); the secondary lines quote the API step as their `value`",
            "api_step": "cholmod_l_factorize(A, L, Common) called again on a matrix with the same pattern (hash of p / i "
                        "checked on every call): H2D of A->x, gather into the resident S on the device, factorization on the "
                        "cached plan, L left in HBM; a new pattern takes the host permutation + full upload again",
            "roofline": roof, "cpu_baseline": cpu,
            "host_seconds": {"generate": t_gen, "analyze": t_analyze, "plan_inside_analyze": t_plan,
                             "first_factorize_incl_plan_h2d": t_first, "analyze_plus_first_factorize": t_analyze + t_first},
        }
        if resid is not None:
            line["residual_2norm"] = resid
            line["factor_checks"] = checks
        if check_note:
            line["residual_2norm_note"] = check_note
        if dist_checks is not None:
            line["factor_checks_distributed"] = dist_checks
        if secondary:
            line["secondary"] = secondary
        if native:
            line["exchange"] = {"backend": "rccl (engine-native, stream-ordered)",
                                "allreduce_calls_per_factorization": int(stats[17]),
                                "allreduce_GB_per_factorization": 1e-9 * stats[18],
                                "all_gather_GB_per_factorization": 1e-9 * stats[25],
                                "all_gather_GB_waited_for_where_issued": 1e-9 * stats[39],
                                "self_test_share_as_world": int(os.environ.get("CHOLMOD_HIP_SHARE_AS_WORLD", "0")) if selftest else None}
        if allreduce is not None:
            nfac = max(args.steps + args.warmup + (0 if args.no_profile_pass else 1), 1)
            line["exchange"] = {"backend": args.dist_backend, "allreduce_calls_per_factorization": allreduce.stats["n"] // nfac,
                                "allreduce_GB_per_factorization": 1e-9 * allreduce.stats["bytes"] / nfac,
                                "allreduce_GB_by_group_size": {str(k): 1e-9 * v / nfac for k, v in sorted(allreduce.stats["by_size"].items())},
                                "self_test_share_as_world": int(os.environ.get("CHOLMOD_HIP_SHARE_AS_WORLD", "0")) if selftest else None}
        sys.stdout.flush()
        with wd._lock:
            already = wd.done
            wd.done = True
        if not already:
            emit(line, real_stdout)
    wd.finish()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
