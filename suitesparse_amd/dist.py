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


def make_allreduce(group=None, device_memory=True):
    """Return a ctypes callback (keep a reference!) implementing the engine's sum
    all-reduce with torch.distributed on `group` (default group if None).
    device_memory=False treats the pointer as host memory (CPU-only tests)."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    calls = {"n": 0, "bytes": 0}

    def _fn(ptr, count, user):
        try:
            calls["n"] += 1
            calls["bytes"] += 8 * int(count)
            if not device_memory:
                buf = (C.c_double * count).from_address(ptr)
                t = torch.from_numpy(np.ctypeslib.as_array(buf))
                dist.all_reduce(t, group=group)
                return 0
            t = torch.as_tensor(_DevView(ptr, count), device="cuda")
            if backend == "nccl":
                dist.all_reduce(t, group=group)
            else:
                h = t.cpu()
                dist.all_reduce(h, group=group)
                t.copy_(h)
            torch.cuda.synchronize()
            return 0
        except Exception as e:          # never unwind through C
            print(f"[suitesparse_amd.dist] all-reduce failed: {e!r}", flush=True)
            return 1

    cb = ch.ALLREDUCE_FN(_fn)
    cb.stats = calls
    return cb
