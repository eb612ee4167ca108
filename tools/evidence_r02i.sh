# Round-2 evidence run on the GPU box: full -m gpu suite, the default bench line, the
# rocprofv3 kernel-trace summary of the same workload, bench lines of the other configs.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -q > $O/r02i_gpu_tests.log 2>&1; tail -3 $O/r02i_gpu_tests.log
python bench.py > $O/r02i_poisson200_bench_line.json 2> $O/r02i_poisson200_bench.err
python bench.py --grid 100 --no-cpu-baseline --steps 5 > $O/r02i_poisson100_bench_line.json 2>/dev/null
python bench.py --workload box3d --grid 42 --no-cpu-baseline --steps 10 > $O/r02i_box42r3_bench_line.json 2>/dev/null
python bench.py --workload poisson2d --grid 1259 --no-cpu-baseline --steps 10 > $O/r02i_poisson2d1259_bench_line.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r02i_stats_p200 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/r02i_poisson200_bench_line_rocprof_run.json 2> $O/r02i_stats_p200.err
rocprofv3 --kernel-trace --stats -d $O/r02i_stats_box42 --output-format csv -- python $R/tools/one_factorization.py --workload box3d --grid 42 --repeat 3 > $O/r02i_stats_box42.log 2>&1
cd $R
for f in r02i_poisson200_bench_line r02i_poisson100_bench_line r02i_box42r3_bench_line r02i_poisson2d1259_bench_line r02i_poisson200_bench_line_rocprof_run; do
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$f", "GF/s %.0f ms %.2f api %.2f resid %.1e updTF %.2f" % (d["value"], d["ms_per_step"], d.get("ms_per_step_api",0), d.get("residual_2norm",-1), r.get("achieved",0)))
except Exception as e: print("$f ERR", e)
PY
done
ls $O/r02i_stats_p200/*/ | head
