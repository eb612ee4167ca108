"""Sum rocprofv3 --pmc counters per kernel.
usage: pmc_by_kernel.py [--second-half] <rocprof_dir> [<rocprof_dir> ...] > out.json
Every *counter_collection.csv below the directories is read; the result maps
kernel (short name) -> {dispatches, counter: sum}.  --second-half keeps, per kernel and
counter, only the later half of its dispatches (a run of one_factorization.py --repeat 1:
the refactorization of the resident matrix, not the first factorization that builds the
assembly map)."""
import collections, csv, glob, json, re, sys

args = sys.argv[1:]
second = "--second-half" in args
args = [a for a in args if a != "--second-half"]
rows = collections.defaultdict(list)
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", n)
            key = (m.group(1) + (m.group(2) or "")) if m else n[:60]
            rows[(key, r["Counter_Name"])].append((int(r.get("Dispatch_Id", r.get("Correlation_Id", "0")) or 0), float(r["Counter_Value"])))
res = collections.defaultdict(dict)
for (key, c), v in rows.items():
    v.sort()
    ids = sorted(set(i for i, _ in v))
    if second:
        keep = set(ids[len(ids) // 2:])
        v = [(i, x) for i, x in v if i in keep]
        ids = sorted(keep)
    res[key][c] = sum(x for _, x in v)
    res[key]["dispatches"] = max(res[key].get("dispatches", 0), len(ids))
print(json.dumps(res, indent=1, sort_keys=True))
