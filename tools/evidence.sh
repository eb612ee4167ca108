#!/bin/bash
# Evidence run on the GPU box, parameterised by a tag (e.g. r03d): the -m gpu suite, the default bench.py
# line (headline 200^3 + the secondary configurations), rocprofv3 --kernel-trace --stats of the same
# command with 1+1 steps, kernel stats of the three secondary workloads, and -- with PMC=1 -- the HBM
# counter passes of the dominant update kernel at the headline size (tools/pmc_top.sh) and of the thin
# stand-in.  usage (from the repo root on the box):  bash tools/evidence.sh <tag> [tests] [PMC=1]
TAG=${1:-r03x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
if [[ " $* " == *" tests "* ]]; then
  python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.log 2>&1; grep -E "passed|failed" $O/${TAG}_gpu_tests.log | tail -2
fi
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_p200 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_line_rocprof_run.json 2> $O/${TAG}_stats_p200.err
for W in "poisson3d 100 p100" "box3d 42 box42r3" "poisson2d 1259 p2d1259"; do set -- $W
  rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_$3 --output-format csv -- python $R/tools/one_factorization.py --workload $1 --grid $2 --repeat 3 > $O/${TAG}_stats_$3.log 2>&1
done
if [[ " $* $PMC " == *"PMC=1"* || "$PMC" == "1" ]]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace -d $O/${TAG}_${C}_p2d --output-format csv -- python $R/tools/one_factorization.py --workload poisson2d --grid 1259 --repeat 1 > $O/${TAG}_${C}_p2d.log 2>&1
  done
  cd $R; python tools/pmc_by_kernel.py --second-half gpurun_out/${TAG}_FETCH_SIZE_p2d gpurun_out/${TAG}_WRITE_SIZE_p2d > gpurun_out/${TAG}_pmc_by_kernel_poisson2d1259.json
fi
cd $R
for d in p200 p100 box42r3 p2d1259; do f=$(ls $O/${TAG}_stats_$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_${d}_kernel_stats.csv && { echo "== $d"; head -7 "$f" | cut -c1-150; }; done
python tools/bench_summary.py $O/${TAG}_bench_line.json $O/${TAG}_bench_line_rocprof_run.json
