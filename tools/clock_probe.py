"""Sample rocm-smi (sclk, power) while the update kernel runs back to back."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), out))
        except Exception as e:
            samples.append((time.time(), repr(e)))
        time.sleep(0.3)
print("idle:", subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-900:])
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
r = ch.probes().cholmod_hip_bench_update_kernel(16384, 16384, 4096, 40, 0)
t1 = time.time()
stop = True; th.join()
print("update kernel", r / 1e12, "TFLOP/s over", t1 - t0, "s")
import json
for t, o in samples[:: max(1, len(samples) // 8)]:
    try:
        d = json.loads(o)
        c = list(d.values())[0]
        print(round(t - t0, 2), {k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "Power" in k})
    except Exception:
        print(round(t - t0, 2), o[:200])
