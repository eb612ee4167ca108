#!/bin/bash
# Counters of EVERY launch of one refactorization of a mid-size workload (round-4 review, item 3): three separate
# rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | matrix-pipe counters), as MI355X_MICROARCH.md prescribes, each over
# `one_factorization.py --repeat 1` (first factorization + one refactorization of the resident matrix; the later half of
# every kernel's dispatches = the refactorization, tools/pmc_by_kernel.py --second-half), summed per kernel.
# usage (repo root, GPU box):  bash tools/pmc_workload.sh <tag> <workload> <grid> <short>
#   e.g.  bash tools/pmc_workload.sh r05c box3d 42 box42r3   ->  gpurun_out/r05c_pmc_by_kernel_box42r3.json
TAG=$1; WL=$2; GRID=$3; SHORT=$4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 1500 rocprofv3 --pmc $C --kernel-trace -d $O/${TAG}_pmc_${SHORT}_$N --output-format csv -- python $R/tools/one_factorization.py --workload $WL --grid $GRID --repeat 1 > $O/${TAG}_pmc_${SHORT}_$N.log 2>&1
  echo "$SHORT $N rc=$?"
done
cd $R
python tools/pmc_by_kernel.py --second-half $O/${TAG}_pmc_${SHORT}_FETCH_SIZE $O/${TAG}_pmc_${SHORT}_WRITE_SIZE $O/${TAG}_pmc_${SHORT}_SQ_VALU_MFMA_BUSY_CYCLES > $O/${TAG}_pmc_by_kernel_${SHORT}.json
rm -rf $O/${TAG}_pmc_${SHORT}_FETCH_SIZE $O/${TAG}_pmc_${SHORT}_WRITE_SIZE $O/${TAG}_pmc_${SHORT}_SQ_VALU_MFMA_BUSY_CYCLES
python - <<PY
import json
d = json.load(open("$O/${TAG}_pmc_by_kernel_${SHORT}.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    tr = 2048.0 * v.get("FETCH_SIZE", 0) + 1024.0 * v.get("WRITE_SIZE", 0)
    act = v.get("GRBM_GUI_ACTIVE", 0)
    util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (act / 8.0 * 1024.0) if act else 0
    print("%-40s n=%5d traffic %9.3f GB  mfma util %5.3f" % (k[:40], v.get("dispatches", 0), tr / 1e9, util))
PY
