"""Summarise a rocprofv3 kernel trace of bench.py: per-stream busy time, overlap,
per-class time on the look-ahead stream, gaps in front of long main-stream kernels."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_assemble' in r['Kernel_Name']]
rows = [r for r in rows[idx[-1]:] if 'peak' not in r['Kernel_Name'] and r['Stream_Id'] != '0']
t0 = int(rows[0]['Start_Timestamp'])
S = lambda r: (int(r['Start_Timestamp']) - t0) / 1e3
E = lambda r: (int(r['End_Timestamp']) - t0) / 1e3
sid = sorted(set(r['Stream_Id'] for r in rows))
main = [r for r in rows if r['Stream_Id'] == sid[0]]
side = [r for r in rows if r['Stream_Id'] != sid[0]]
ev = sorted((S(r), E(r)) for r in rows)
cov, (cs, ce) = 0, ev[0]
for s, e in ev[1:]:
    if s > ce:
        cov += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
cov += ce - cs
mb, sb = sum(E(r) - S(r) for r in main), sum(E(r) - S(r) for r in side)
print(f"span {max(E(r) for r in rows)/1e3:.2f} ms  main busy {mb/1e3:.2f}  side busy {sb/1e3:.2f}  union {cov/1e3:.2f}  overlap {(mb+sb-cov)/1e3:.2f}")
for name, grp in (("main", main), ("side", side)):
    c = collections.Counter(); n = collections.Counter()
    for r in grp:
        k = [x for x in ('k_potrf', 'k_trsm', 'k_update2', 'k_extend_add', 'k_zero', 'k_assemble', 'fillBuffer') if x in r['Kernel_Name']]
        k = k[0] if k else r['Kernel_Name'][:16]
        c[k] += E(r) - S(r); n[k] += 1
    print(name, {k: f"{v/1e3:.2f} ms / {n[k]}" for k, v in c.items()})
prev, gaps = None, 0
for r in main:
    if prev is not None and S(r) - prev > 20:
        gaps += S(r) - prev
    prev = E(r)
print(f"main-stream idle gaps > 20 us: {gaps/1e3:.2f} ms")
