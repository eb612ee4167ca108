#!/bin/bash
# round-5 A/B batch (GPU box, repo root): un-profiled ms per step of the three secondary workloads under the chain / outer-block knobs
R=$GRAFT_REPO_ROOT; cd $R
bash tools/ab.sh CHOLMOD_HIP_CHAINF_AUTO "0 1"
echo "--- box42: outer block thresholds (rows from which a front takes OB 1024 / 2048)"
for T in "4000:8000" "4000:6000" "3000:6000" "2500:5000" "6000:12000"; do
  ms=$(env CHOLMOD_HIP_OB1024_ROWS=${T%%:*} CHOLMOD_HIP_OB2048_ROWS=${T##*:} python bench.py --workload box3d --grid 42 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms (api %.3f)' % (d['ms_per_step'], d['ms_per_step_api']))")
  echo "box3d 42 OB1024_ROWS:OB2048_ROWS=$T  $ms"
  ms=$(env CHOLMOD_HIP_CHAINF_AUTO=1 CHOLMOD_HIP_OB1024_ROWS=${T%%:*} CHOLMOD_HIP_OB2048_ROWS=${T##*:} python bench.py --workload box3d --grid 42 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms (api %.3f)' % (d['ms_per_step'], d['ms_per_step_api']))")
  echo "box3d 42 CHAINF_AUTO=1 OB1024_ROWS:OB2048_ROWS=$T  $ms"
done
