"""Sum rocprofv3 --pmc counters per kernel.
usage: pmc_by_kernel.py <rocprof_dir> [<rocprof_dir> ...] > out.json
Every *counter_collection.csv below the directories is read; the result maps
kernel (short name) -> {dispatches, counter: sum}."""
import collections, csv, glob, json, re, sys

out = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", n)
            key = (m.group(1) + (m.group(2) or "")) if m else n[:60]
            out[key][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(key, r["Counter_Name"])].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
res = {}
for k, v in out.items():
    res[k] = dict(v)
    res[k]["dispatches"] = max(len(disp[(k, c)]) for c in v)
print(json.dumps(res, indent=1, sort_keys=True))
