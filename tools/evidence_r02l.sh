# Round-2 evidence run of the last commit (same commands as evidence_r02j.sh) on the GPU box: -m gpu suite, bench lines of the four
# configurations, rocprofv3 kernel-trace summaries, PMC passes of the thin-regime stand-in.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -q > $O/r02l_gpu_tests.log 2>&1; grep -E "passed|failed" $O/r02l_gpu_tests.log | tail -2
python bench.py > $O/r02l_poisson200_bench_line.json 2> $O/r02l_poisson200_bench.err
python bench.py --grid 100 --no-cpu-baseline --steps 5 > $O/r02l_poisson100_bench_line.json 2>/dev/null
python bench.py --workload box3d --grid 42 --no-cpu-baseline --steps 10 > $O/r02l_box42r3_bench_line.json 2>/dev/null
python bench.py --workload poisson2d --grid 1259 --no-cpu-baseline --steps 10 > $O/r02l_poisson2d1259_bench_line.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r02l_stats_p200 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/r02l_poisson200_bench_line_rocprof_run.json 2> $O/r02l_stats_p200.err
rocprofv3 --kernel-trace --stats -d $O/r02l_stats_box42 --output-format csv -- python $R/tools/one_factorization.py --workload box3d --grid 42 --repeat 3 > $O/r02l_stats_box42.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r02l_stats_p2d --output-format csv -- python $R/tools/one_factorization.py --workload poisson2d --grid 1259 --repeat 3 > $O/r02l_stats_p2d.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r02l_fetch_p2d --output-format csv -- python $R/tools/one_factorization.py --workload poisson2d --grid 1259 --repeat 1 > $O/r02l_fetch_p2d.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r02l_write_p2d --output-format csv -- python $R/tools/one_factorization.py --workload poisson2d --grid 1259 --repeat 1 > $O/r02l_write_p2d.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $O/r02l_insts_p2d --output-format csv -- python $R/tools/one_factorization.py --workload poisson2d --grid 1259 --repeat 1 > $O/r02l_insts_p2d.log 2>&1
cd $R
python tools/pmc_by_kernel.py --second-half gpurun_out/r02l_fetch_p2d gpurun_out/r02l_write_p2d > gpurun_out/r02l_pmc_by_kernel_poisson2d1259.json
python tools/pmc_by_kernel.py --second-half gpurun_out/r02l_insts_p2d > gpurun_out/r02l_pmc_insts_thin_poisson2d1259.json
for f in r02l_poisson200_bench_line r02l_poisson100_bench_line r02l_box42r3_bench_line r02l_poisson2d1259_bench_line r02l_poisson200_bench_line_rocprof_run; do
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$f", "GF/s %.0f ms %.2f api %.2f resid %.1e updTF %.2f" % (d["value"], d["ms_per_step"], d.get("ms_per_step_api",0), d.get("residual_2norm",-1), r.get("achieved",0)), r.get("thin_front_kernel"))
except Exception as e: print("$f ERR", e)
PY
done
for d in r02l_stats_p200 r02l_stats_box42 r02l_stats_p2d; do f=$(ls $O/$d/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== $d $f"; head -8 "$f" | cut -c1-160; done
grep -A8 "k_thin_front" gpurun_out/r02l_pmc_by_kernel_poisson2d1259.json | head -40
