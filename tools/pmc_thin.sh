#!/bin/bash
# HBM-side traffic of the thin-front kernels on the G3_circuit stand-in (2D Poisson 1259^2): FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 --pmc passes over analyze + factorization + ONE refactorization; pmc_by_kernel.py --second-half keeps the
# refactorization (assembly map in place, leaf fronts two to a wave).  usage (repo root, GPU box): bash tools/pmc_thin.sh <tag>
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/${TAG}_${C}_p2d --output-format csv -- python $R/tools/one_factorization.py --workload poisson2d --grid 1259 --repeat 1 > $O/${TAG}_${C}_p2d.log 2>&1
  echo "$C rc=$?"
done
cd $R; python tools/pmc_by_kernel.py --second-half gpurun_out/${TAG}_FETCH_SIZE_p2d gpurun_out/${TAG}_WRITE_SIZE_p2d > gpurun_out/${TAG}_pmc_by_kernel_poisson2d1259.json
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_pmc_by_kernel_poisson2d1259.json'))
f=sum(v.get('FETCH_SIZE',0) for k,v in d.items() if k.startswith(('k_thin_front','k_leaf_pair'))); w=sum(v.get('WRITE_SIZE',0) for k,v in d.items() if k.startswith(('k_thin_front','k_leaf_pair')))
print('thin kernels: fetched %.3f GB (x2) written %.3f GB' % (2*1024*f/1e9, 1024*w/1e9))"
