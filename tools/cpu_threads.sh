#!/bin/bash
# the product's CPU path (Common->useGPU = 0) on Poisson m^3 at several thread counts and grains (Mflop of a dense call per BLAS thread)
R=${GRAFT_REPO_ROOT:-.}; M=${1:-100}
B=$(python -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench._scipy_openblas())")
for G in ${2:-32}; do for T in ${3:-16 32 64}; do
  echo -n "grain $G Mflop/thread, $T threads: "
  CHOLMOD_CPU_MFLOP_PER_THREAD=$G OMP_NUM_THREADS=$T OPENBLAS_NUM_THREADS=$T BENCH_ROOT=$R BENCH_CPU_M=$M BENCH_CPU_REPS=1 CHOLMOD_BLAS_LIBRARY=$B python -c "import sys; sys.path.insert(0,'$R'); import bench, json, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf): exec(bench.CPU_CHILD)
r = json.loads(buf.getvalue().strip().splitlines()[-1]); print('%.1f GFLOP/s (%.2f s)' % (r['fl'] / min(r['seconds']) / 1e9, min(r['seconds'])))"
done; done
