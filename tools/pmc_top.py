"""Helper of tools/pmc_top.sh.
  select <trace_dir> <kernel-substring> <ntop> <out.json>  -> prints the --kernel-iteration-range arguments
  summarise <selection.json> <fetch_dir> <write_dir> <workload> <out.json>
Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes
(MI355X_MICROARCH.md, HBM section; calibrated here on k_factor_checks): traffic = 2 x FETCH + WRITE."""
import csv, glob, json, sys


def trace_rows(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    return list(csv.DictReader(open(f)))


if sys.argv[1] == "select":
    d, sub, ntop, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    rows = [r for r in trace_rows(d) if sub in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]     # ms
    order = sorted(range(len(rows)), key=lambda i: -dur[i])[:ntop]
    sel = sorted(order)
    json.dump({"kernel_substring": sub, "launches_of_this_kernel": len(rows), "iterations": [i + 1 for i in sel],
               "ms": [dur[i] for i in sel], "share_of_kernel_time": sum(dur[i] for i in sel) / max(sum(dur), 1e-30),
               "kernel_ms_total": sum(dur)}, open(out, "w"))
    # (rocprofv3 counts kernel iterations from 1)
    print(" ".join("[%d-%d]" % (i + 1, i + 1) for i in sel))
else:
    selj, fd, wd, workload, out = sys.argv[2:7]
    md = sys.argv[7] if len(sys.argv) > 7 else None
    sel = json.load(open(selj))

    def total(d, counter):
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
        n, s = 0, 0.0
        seen = set()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or sel["kernel_substring"] not in r["Kernel_Name"]:
                continue
            s += float(r["Counter_Value"])
            seen.add(r.get("Dispatch_Id"))
        return len(seen), s
    nf, sf = total(fd, "FETCH_SIZE")
    nw, sw = total(wd, "WRITE_SIZE")
    fetch = 2.0 * 1024.0 * sf / max(nf, 1)
    write = 1024.0 * sw / max(nw, 1)
    res = {"workload": workload, "kernel": sel["kernel_substring"], "kernel_kind": 12 if "update3" in sel["kernel_substring"] else 5,
           "launches_profiled": nf, "launches_profiled_write_pass": nw,
           "selection": "the %d longest of the %d launches of %s in one factorization (%.0f %% of its time), "
                        "rocprofv3 --kernel-iteration-range" % (len(sel["iterations"]), sel["launches_of_this_kernel"],
                                                                sel["kernel_substring"], 100 * sel["share_of_kernel_time"]),
           "ms_per_launch_in_the_trace_pass": sum(sel["ms"]) / max(len(sel["ms"]), 1),
           "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
           "calibration": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950 counts a 128-byte request as 64), WRITE_SIZE (KiB) x 1024",
           "command": "tools/pmc_top.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-include-regex ... --kernel-iteration-range ... "
                      "--kernel-trace -- python tools/one_factorization.py --grid 200 (separate passes)"}
    if md and glob.glob(md + "/**/*counter_collection.csv", recursive=True):
        # matrix-pipe utilisation of the same launches: MFMA busy cycles / (GPU-active cycles per XCD x 1024 SIMDs)
        # (GRBM_GUI_ACTIVE is reported per XCD and summed over the 8 of them; 256 CUs x 4 SIMDs)
        f = glob.glob(md + "/**/*counter_collection.csv", recursive=True)[0]
        acc, ids = {}, set()
        for r in csv.DictReader(open(f)):
            if sel["kernel_substring"] not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            ids.add(r.get("Dispatch_Id"))
        tr = [r for r in trace_rows(md) if sel["kernel_substring"] in r["Kernel_Name"]]
        secs = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr) * 1e-9
        busy, act, mops = acc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), acc.get("GRBM_GUI_ACTIVE", 0.0), acc.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
        if act > 0:
            res["mfma_utilisation"] = {
                "launches": len(ids), "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_over_8_XCDs": act,
                "SQ_INSTS_VALU_MFMA_MOPS_F64": mops, "kernel_seconds_in_this_pass": secs,
                "mfma_pipe_utilisation": busy / (act / 8.0 * 1024.0),
                "shader_clock_GHz_under_the_counters": (act / 8.0 / secs / 1e9) if secs > 0 else None,
                "note": "utilisation = MFMA busy cycles / (GPU-active cycles per XCD x 1024 SIMDs); a separate rocprofv3 --pmc pass over the same launches"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))
