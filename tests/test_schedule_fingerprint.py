"""The engine's host-side plans, pinned: fingerprints (cholmod_hip_debug_schedule_hash: fronts, routing, layout, every
group array, the launch list) of a battery of problems x worlds x ranks x plan flags x tuning knobs, recorded in
tests/golden/schedule_fingerprints.json by tools/schedule_fingerprints.py.  A restructuring of plan_build.hip /
schedule_dense.hip must not move a single launch; a deliberate change of the schedule re-records them (`write`) in the
same commit.  No GPU needed: host-only plans."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_schedules_match_the_recorded_fingerprints():
    import schedule_fingerprints as F
    ref = json.load(open(F.OUT))
    fp = F.fingerprints()
    assert set(fp) == set(ref)
    bad = [k for k in ref if fp[k] != ref[k]]
    assert not bad, (len(bad), bad[:5])
    assert len({tuple(v) for v in fp.values()}) > 400        # (the battery does tell its cases apart)


def test_product_library_has_no_test_hooks():
    """CHOLMOD_HIP_TEST_* (jitter, poison, dropped waits, injected failures, a hung exchange) are compiled into
    lib/libcholmod_amd_testhooks.so only: the names do not occur in the product library, so no environment variable can
    make it compute a wrong factor or fail on purpose."""
    from suitesparse_amd import cholmod as ch
    prod = open(ch.LIB_PATH, "rb").read()
    assert b"CHOLMOD_HIP_TEST_" not in prod and b"_TEST_JITTER" not in prod
    hooks = open(ch.HOOKS_LIB_PATH, "rb").read()
    for name in (b"CHOLMOD_HIP_TEST_JITTER", b"CHOLMOD_HIP_TEST_DROP_WAITS", b"CHOLMOD_HIP_TEST_POISON_ARENA",
                 b"CHOLMOD_HIP_TEST_FAIL_LAUNCH", b"CHOLMOD_HIP_TEST_HANG_EXCHANGE"):
        assert name in hooks, name
    lib = ch.lib(hooks=True)
    for name in ch.API_SYMBOLS + ch.HIP_SYMBOLS:
        assert hasattr(lib, name), name


def test_no_function_of_the_engine_sources_exceeds_300_lines():
    """Round-4 review: schedule_dense (870 lines) and build_host (680) carried every concern in one body.  They are now
    classes with one method per concern (schedule_dense.hip, plan_build.hip); this keeps them that way."""
    import re
    hip = os.path.join(ROOT, "suitesparse_amd", "csrc", "hip")
    for fn in ("engine.hip", "plan_build.hip", "schedule_dense.hip"):
        lines = open(os.path.join(hip, fn)).read().split("\n")
        # a function body = a line that opens at column 0 or 4 with '{' after a signature line, to its matching close
        for indent in ("", "    "):
            start = None
            for i, ln in enumerate(lines):
                if ln == indent + "{" and i > 0 and re.search(r"\)\s*(const)?\s*$", lines[i - 1]) and start is None:
                    start = i
                elif ln in (indent + "}", indent + "} ;") and start is not None:
                    assert i - start <= 300, (fn, start + 1, i - start)
                    start = None
