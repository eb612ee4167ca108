#!/bin/bash
# Round 5: A/B of CHOLMOD_HIP_UPDW_ALONE_TILES (a region of that many tiles is a k_update3 launch of its own) at the headline
# size and on the mid-size workloads, un-profiled.  usage (repo root, GPU box): bash tools/r05_alone.sh "0 32768 4096 2048"
R=${GRAFT_REPO_ROOT:-.}; VALS=${1:-"0 8192 32768 131072"}
for v in $VALS; do
  CHOLMOD_HIP_UPDW_ALONE_TILES=$v python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('200^3 ALONE_TILES=$v: %.1f ms  %.2f TF resid %.1e launches %d' % (d['ms_per_step'], d['value']/1e3, d['residual_2norm'], d['config']['launches_per_step']))"
done
bash $R/tools/ab.sh CHOLMOD_HIP_UPDW_ALONE_TILES "$VALS"
