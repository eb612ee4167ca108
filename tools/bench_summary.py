"""Compact summary of bench.py lines: of the JSON files given, or -- with bench.py arguments instead
of files -- of a bench.py run started here.  Headline, per-class seconds, secondary workloads."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def show(d, tag=""):
    r = d.get("roofline") or {}
    print(tag, d["config"]["workload"][:24],
          "| GF/s %.0f ms %.2f api %.2f resid %s launches %d" % (d["value"], d["ms_per_step"], d.get("ms_per_step_api", 0),
          ("%.1e" % d["residual_2norm"]) if "residual_2norm" in d else "-", d["config"]["launches_per_step"]),
          "| %s %.1f TF" % ((r.get("kernel") or "")[:12], r.get("achieved", 0)),
          {k: round(v * 1e3, 2) for k, v in r.get("seconds_by_class", {}).items()}, r.get("thin_front_kernel"))
    if d.get("cpu_baseline"):
        print("   cpu_baseline", d["cpu_baseline"])
    if d.get("measured_fp64_mfma_ceiling_TFLOPs"):
        print("   mfma ceiling %.1f TF, update kernel 16384^2 x 4096: %s" % (d["measured_fp64_mfma_ceiling_TFLOPs"],
              d.get("measured_update_kernel_TFLOPs_16384x16384x4096")))
    for s in d.get("secondary", []):
        if "complex" in s:
            print("   secondary", s["workload"][:40], "| real %.2f ms, complex %.2f ms (x%.2f), residuals %.1e / %.1e" % (
                s["real"]["ms_per_step"], s["complex"]["ms_per_step"], s["complex_over_real_time"],
                s["real"]["residual_2norm"], s["complex"]["residual_2norm"]))
            continue
        rr = s.get("roofline") or {}
        print("   secondary", s.get("workload", "?")[:28], "| GF/s %.0f (%.1f %%) ms %.3f api %.3f resid %.1e" % (
            s.get("value", 0), s.get("pct_fp64_mfma_peak", 0), s.get("ms_per_step", 0), s.get("ms_per_step_api", 0),
            s.get("residual_2norm", -1)), "| %.1f TF" % rr.get("achieved", 0),
            {k: round(v * 1e3, 3) for k, v in rr.get("seconds_by_class", {}).items()}, s.get("hbm_roofline"), s.get("error"))


files = [a for a in sys.argv[1:] if a.endswith(".json") and os.path.exists(a)]
if files:
    for f in files:
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            show(json.loads(lines[-1]), os.path.basename(f))
        else:
            print(f, "NO JSON")
else:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print("NO JSON", out.stdout[-2000:], out.stderr[-3000:])
        sys.exit(1)
    show(json.loads(lines[-1]), " ".join(sys.argv[1:]))
