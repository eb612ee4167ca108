#!/bin/bash
# half tiles A/B (GPU box): un-profiled ms per step, secondary workloads and the headline
R=$GRAFT_REPO_ROOT; cd $R
bash tools/ab.sh CHOLMOD_HIP_UPD3_HALF_MAX "0 10240 6144 16384"
# (the pooling threshold CHOLMOD_HIP_UPD3_HALF_MIN "256 1024" was part of this batch; flat, a constant since)
for v in 0 10240; do
  CHOLMOD_HIP_UPD3_HALF_MAX=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('200^3 HALF_MAX=$v: %.1f ms  %.2f TF resid %.1e' % (d['ms_per_step'], d['value']/1e3, d['residual_2norm']))"
done
