"""Timeline of the last kernels of a rocprofv3 --kernel-trace run: start / end (us), queue, kernel, grid.
usage: trace_tail.py <kernel_trace.csv> [n=400]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-n:]:
    print("%10.1f %10.1f q%s %-34s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"),
          r["Kernel_Name"].replace("void sship::", "").replace("sship::", "")[:34], r.get("Grid_Size_X", "?")))
