# final evidence refresh of the session: -m gpu suite, bench lines, what-if per-rank compute
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -q > $O/r02m_gpu_tests.log 2>&1; grep -E "passed|failed" $O/r02m_gpu_tests.log | tail -2
python bench.py > $O/r02m_poisson200_bench_line.json 2> $O/r02m_poisson200_bench.err
python bench.py --grid 100 --no-cpu-baseline --steps 5 > $O/r02m_poisson100_bench_line.json 2>/dev/null
python bench.py --workload box3d --grid 42 --no-cpu-baseline --steps 10 > $O/r02m_box42r3_bench_line.json 2>/dev/null
python bench.py --workload poisson2d --grid 1259 --no-cpu-baseline --steps 10 > $O/r02m_poisson2d1259_bench_line.json 2>/dev/null
for f in r02m_poisson200_bench_line r02m_poisson100_bench_line r02m_box42r3_bench_line r02m_poisson2d1259_bench_line; do
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$f", "GF/s %.0f ms %.2f api %.2f resid %.1e updTF %.2f solve_ms %.2f" % (d["value"], d["ms_per_step"], d.get("ms_per_step_api",0), d.get("residual_2norm",-1), r.get("achieved",0), d.get("factor_checks",{}).get("solve_device_ms",-1)), r.get("thin_front_kernel"))
except Exception as e: print("$f ERR", e)
PY
done
