#!/bin/bash
# resident and API step of the three mid-size workloads under one environment variable's values (un-profiled A/B)
# usage: bash tools/ab_resident.sh VAR "v1 v2 ..."   (the value "-" = variable unset)
R=${GRAFT_REPO_ROOT:-.}; VAR=$1; shift
for v in $1; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  for W in "poisson2d 1259" "box3d 42" "poisson3d 100"; do
    echo "$VAR=$v  $(timeout 200 python $R/tools/api_probe.py $W 2>&1 | tail -1)"
  done
done
