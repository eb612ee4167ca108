#!/bin/bash
# Round 5: the outer-block row thresholds once more, on the final kernel (defaults 4000 : 8000 : 24000 rows for 1024 : 2048 : 4096 columns)
R=${GRAFT_REPO_ROOT:-.}
for v in "4000 8000 24000" "4000 8000 16000" "3000 6000 16000" "4000 8000 32000"; do set -- $v
  CHOLMOD_HIP_OB1024_ROWS=$1 CHOLMOD_HIP_OB2048_ROWS=$2 CHOLMOD_HIP_OB4096_ROWS=$3 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('200^3 OB rows $1:$2:$3  %.1f ms  %.2f TF resid %.1e launches %d' % (d['ms_per_step'], d['value']/1e3, d['residual_2norm'], d['config']['launches_per_step']))"
done
for v in "4000 8000 24000" "3000 6000 16000" "3000 6000 12000"; do set -- $v
  CHOLMOD_HIP_OB1024_ROWS=$1 CHOLMOD_HIP_OB2048_ROWS=$2 CHOLMOD_HIP_OB4096_ROWS=$3 python $R/bench.py --workload poisson3d --grid 100 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('100^3 OB rows $1:$2:$3  %.3f ms  %.2f TF' % (d['ms_per_step'], d['value']/1e3))"
done
