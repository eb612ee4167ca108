#!/bin/bash
# resident and API step of the three mid-size workloads with two builds of the product library (un-profiled A/B):
# usage: bash tools/ab_lib.sh <other libcholmod_amd.so> [rounds=2]     (the other build against the one in lib/)
R=${GRAFT_REPO_ROOT:-.}; OTHER=$1; N=${2:-2}
for r in $(seq $N); do
  for L in "$OTHER" ""; do
    if [ -n "$L" ]; then export CHOLMOD_AMD_LIB=$L; tag=other; else unset CHOLMOD_AMD_LIB; tag=this; fi
    for W in "poisson2d 1259" "box3d 42" "poisson3d 100"; do
      echo "$tag  $(timeout 200 python $R/tools/api_probe.py $W 2>&1 | tail -1)"
    done
  done
done
