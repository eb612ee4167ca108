#!/usr/bin/env python3
"""Time line of a `rocprofv3 --kernel-trace` run of tools/one_factorization.py: per factorization (split at the
hipMemset of Lx / the k_assemble* kernels) the wall time from first kernel start to last kernel end, the busy time (union
of kernel intervals), the idle gaps between consecutive kernels, and per kernel name count / total / mean / min.
usage: trace_gaps.py <rocprof_dir> [factorization-index=-1]"""
import csv, glob, re, sys, collections

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


starts = [i for i, r in enumerate(rows) if "k_assemble" in r["Kernel_Name"]]
if not starts:
    starts = [0]
starts.append(len(rows))
segs = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
a, b = segs[which]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in seg)
busy, cur_s, cur_e = 0, None, None
gaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, short(r["Kernel_Name"])))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
print("factorization %d of %d: %d kernels, wall %.3f ms, busy %.3f ms, idle %.3f ms in %d gaps (mean %.2f us, max %.1f us)" % (
    which, len(segs), len(seg), (t1 - t0) * 1e-6, busy * 1e-6, (t1 - t0 - busy) * 1e-6, len(gaps),
    (sum(g for g, _ in gaps) / max(len(gaps), 1)) * 1e-3, max([g for g, _ in gaps] or [0]) * 1e-3))
per = collections.defaultdict(list)
for r in seg:
    per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("%-44s %6s %10s %9s %9s" % ("kernel", "n", "total ms", "mean us", "min us"))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("%-44s %6d %10.3f %9.1f %9.1f" % (k, len(v), sum(v) * 1e-3, sum(v) / len(v), min(v)))
gk = collections.defaultdict(list)
for g, k in gaps:
    gk[k].append(g * 1e-3)
print("idle before a kernel of kind:")
for k, v in sorted(gk.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("  %-42s %6d gaps %9.3f ms  mean %7.2f us" % (k, len(v), sum(v) * 1e-3, sum(v) / len(v)))
