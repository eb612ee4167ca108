"""Complex (Hermitian) Poisson m^3: the twin with even-column updates against the plain embedding and
the real factorization of the same pattern (bench.py: complex_line).  usage: complex_time.py [m ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for m in [int(a) for a in sys.argv[1:]] or [64]:
    print(json.dumps(bench.complex_line(m)), flush=True)
