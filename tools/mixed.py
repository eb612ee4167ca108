import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import cholmod as ch
L = ch.lib()
for bpc in (2, 4):
    for (im, iv) in [(20000, 0), (0, 220000), (20000, 220000), (20000, 110000), (20000, 440000)]:
        t = ch.probes().cholmod_hip_bench_mixed(bpc, im, iv)
        blocks = 256 * bpc
        fm = blocks * 2 * im * 8 * 2048.0
        fv = blocks * 2 * 64 * iv * 16 * 2.0
        print(f"blocks/CU {bpc} it_mfma {im} it_valu {iv}: {t*1e3:.2f} ms  mfma {fm/t/1e12:.2f} TF + valu {fv/t/1e12:.2f} TF = {(fm+fv)/t/1e12:.2f} TF", flush=True)
