#!/bin/bash
# Round 5: A/B of CHOLMOD_HIP_SWZ16 (16 x 16 super-tiles per XCD) now that partial tiles no longer scatter the waves.
R=${GRAFT_REPO_ROOT:-.}
for v in 0 1; do
  if [ $v = 1 ]; then export CHOLMOD_HIP_SWZ16=1; else unset CHOLMOD_HIP_SWZ16; fi
  python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('200^3 SWZ16=$v: %.1f ms  %.2f TF resid %.1e' % (d['ms_per_step'], d['value']/1e3, d['residual_2norm']))"
  for W in "box3d 42 10" "poisson3d 100 5"; do set -- $W
    python $R/bench.py --workload $1 --grid $2 --steps $3 --warmup 2 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 SWZ16=$v: %.3f ms  %.2f TF' % (d['ms_per_step'], d['value']/1e3))"
  done
done
