#!/bin/bash
# resident step against API step (cholmod_l_factorize from host memory) of the three secondary workloads
R=${GRAFT_REPO_ROOT:-.}
for W in "poisson2d 1259 20" "box3d 42 10" "poisson3d 100 5"; do set -- $W
  python $R/bench.py --workload $1 --grid $2 --steps $3 --warmup 2 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2: resident %.3f ms  api %.3f ms  (+%.3f)' % (d['ms_per_step'], d['ms_per_step_api'], d['ms_per_step_api'] - d['ms_per_step']))"
done
