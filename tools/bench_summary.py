"""Run bench.py with the given args and print a compact summary line."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not lines:
    print("NO JSON", out.stdout[-2000:], out.stderr[-3000:])
    sys.exit(1)
d = json.loads(lines[-1])
r = d.get("roofline") or {}
print(d["config"]["workload"][:16], " ".join(sys.argv[1:]),
      "| GF/s %.0f ms %.2f dev_ms %.2f resid %s launches %d" % (d["value"], d["ms_per_step"], d["device_ms_per_step"],
      ("%.1e" % d["residual_2norm"]) if "residual_2norm" in d else "-", d["config"]["launches_per_step"]),
      "| upd TF %.1f" % r.get("achieved", 0),
      {k: round(v * 1e3, 2) for k, v in r.get("seconds_by_class", {}).items()}, r.get("thin_front_kernel"))
