#!/bin/bash
# Outer-block thresholds (rows from which a front takes OB = 1024 / 2048 / 4096) at a given grid: ms per step for a list of
# "t1:t2:t3" settings.  usage: bash tools/ob_sweep.sh <grid> "4000:8000:24000 2000:8000:24000 ..."
GRID=$1; R=${GRAFT_REPO_ROOT:-.}
for s in $2; do IFS=: read a b c <<< "$s"
  ms=$(env CHOLMOD_HIP_OB1024_ROWS=$a CHOLMOD_HIP_OB2048_ROWS=$b CHOLMOD_HIP_OB4096_ROWS=$c python $R/bench.py --grid $GRID --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-profile-pass --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms  %.2f TF' % (d['ms_per_step'], d['value']/1e3))")
  echo "grid $GRID  OB1024/2048/4096 from $a/$b/$c rows:  $ms"
done
