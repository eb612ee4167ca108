#!/bin/bash
# A/B of an environment toggle on the three secondary workloads: bash tools/ab.sh VAR "valA valB" -> ms per step
VAR=$1; VALS=$2; R=${GRAFT_REPO_ROOT:-.}
for W in "poisson2d 1259 10" "box3d 42 10" "poisson3d 100 5"; do set -- $W
  for v in $VALS; do
    ms=$(env $VAR=$v python $R/bench.py --workload $1 --grid $2 --steps $3 --warmup 2 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.2f TF  resid %.1e' % (d['ms_per_step'], d['value']/1e3, d['residual_2norm']))")
    echo "$1 $2  $VAR=$v  $ms"
  done
done
