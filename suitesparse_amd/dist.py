"""torch.distributed plumbing for the multi-GPU engine (one process per GPU).

The engine's only exchange is an in-place sum all-reduce of a device range
(include/cholmod_hip.h, cholmod_hip_allreduce_fn).  This module provides that
callback on top of torch.distributed:

* backend "nccl" (= RCCL over xGMI on ROCm): the device range is wrapped
  zero-copy as a torch tensor and reduced in place;
* backend "gloo" (CPU tests, or several ranks sharing one GPU in the parity
  tests): staged through host memory.

Nothing here computes; PyTorch is used for the collective only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import cholmod as ch


class _DevView:
    """Expose a raw device pointer through __cuda_array_interface__."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {
            "shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
            "version": 2, "strides": None}


MAX_SUBGROUP_WORLD = 16


def make_allreduce(group=None, device_memory=True, subgroups="all"):
    """Return a ctypes callback (keep a reference!) implementing the engine's sum
    all-reduce with torch.distributed.  The engine names the contiguous rank
    range [first, first+size) that takes part: the whole world uses `group`
    (default group if None); every proper sub-range of size >= 2 needs its own
    process group, created collectively and in the same order on every rank:
    subgroups="all" creates all of them here (fine for a handful of ranks);
    subgroups=None creates none -- call cb.create_groups(ranges) with the ranges
    the plan actually uses (cholmod_hip_get_groups; identical on every rank)
    before the first factorization.
    device_memory=False treats the pointer as host memory (CPU-only tests)."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    calls = {"n": 0, "bytes": 0, "by_size": {}}
    side = [None]
    groups = {(0, world): group}
    base = list(range(world)) if group is None else dist.get_process_group_ranks(group)

    def create_groups(ranges):
        """Collective: every rank passes the same list of (first, size)."""
        for first, size in sorted({(int(a), int(b)) for a, b in ranges}):
            if size < 2 or size >= world or (first, size) in groups:
                continue
            g = dist.new_group([base[r] for r in range(first, first + size)], backend=backend)
            # (ranks outside the range keep the handle too; they never use it)
            groups[(first, size)] = g

    if subgroups == "all" and 2 < world <= MAX_SUBGROUP_WORLD:
        create_groups([(first, size) for size in range(world - 1, 1, -1)
                       for first in range(0, world - size + 1)])

    def _fn(ptr, count, first, size, user):
        try:
            calls["n"] += 1
            calls["bytes"] += 8 * int(count)
            calls["by_size"][int(size)] = calls["by_size"].get(int(size), 0) + 8 * int(count)
            if size <= 1 and world > 1:
                return 0
            key = (int(first), int(size)) if world > 1 else (0, world)
            if key not in groups:
                raise RuntimeError(f"no process group for ranks [{first}, {first + size}) on rank {rank} "
                                   f"(world {world}): create it with cb.create_groups(...) on every rank, or "
                                   f"set CHOLMOD_HIP_NO_SUBGROUPS=1")
            g = groups[key]
            if not device_memory:
                buf = (C.c_double * count).from_address(ptr)
                t = torch.from_numpy(np.ctypeslib.as_array(buf))
                dist.all_reduce(t, group=g)
                return 0
            # a private non-blocking stream: the engine may still be running trailing
            # updates on its own stream (exchange look-ahead), and anything issued on
            # the legacy default stream would wait for them
            if side[0] is None:
                side[0] = torch.cuda.Stream()
            with torch.cuda.stream(side[0]):
                t = torch.as_tensor(_DevView(ptr, count), device="cuda")
                if backend == "nccl":
                    dist.all_reduce(t, group=g)
                else:
                    h = t.cpu()
                    dist.all_reduce(h, group=g)
                    t.copy_(h)
            side[0].synchronize()
            return 0
        except Exception as e:          # never unwind through C
            print(f"[suitesparse_amd.dist] all-reduce failed: {e!r}", flush=True)
            return 1

    cb = ch.ALLREDUCE_FN(_fn)
    cb.stats = calls
    cb.create_groups = create_groups
    return cb
