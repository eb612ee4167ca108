#!/bin/bash
# Round 5, last step: partial tiles that load like whole ones + squares ahead of trapezoids in a k_update3 launch.
# Agreement of the kernel forms, the standalone shapes, then the un-profiled A/B.  usage (repo root, GPU box): bash tools/r05_lockstep.sh
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out
python $R/tools/upd3.py tail 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail diff max', max(d['diff'].values()))"
python $R/tools/upd3.py half 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('half diff max', max(d['diff'].values()))"
python $R/tools/upd3.py standalone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('standalone diff max', max(d['diff'].values()))"
python $R/tools/upd3.py trap 2>/dev/null
python $R/tools/upd3.py pair 2>/dev/null
for v in 0 1; do
  CHOLMOD_HIP_UPDW_SQUARES_FIRST=$v python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('200^3 SQUARES_FIRST=$v: %.1f ms  %.2f TF resid %.1e launches %d' % (d['ms_per_step'], d['value']/1e3, d['residual_2norm'], d['config']['launches_per_step']))"
done
bash $R/tools/ab.sh CHOLMOD_HIP_UPDW_SQUARES_FIRST "0 1"
