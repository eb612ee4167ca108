#!/bin/bash
# HBM-side traffic of the dominant update kernel at the headline size (Poisson 200^3): FETCH_SIZE and
# WRITE_SIZE (separate rocprofv3 --pmc passes, as MI355X_MICROARCH.md prescribes) on the NTOP longest
# launches of ONE factorization -- a counter pass over every launch of a 200^3 factorization does not
# finish (rounds 1 and 2).  Pass 0 is a plain --kernel-trace of the same command: it yields the
# per-kernel iteration index and the duration of every launch, i.e. the selection.
# usage (repo root, on the GPU box):  bash tools/pmc_top.sh <tag> [grid=200] [regex='k_update3<4, 0, 1>'] [ntop=48]
TAG=${1:-r03}; GRID=${2:-200}; RE=${3:-k_update3<4, 0, 1>}; NTOP=${4:-48}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/${TAG}_pmctrace --output-format csv -- python $R/tools/one_factorization.py --grid $GRID > $O/${TAG}_pmctrace.log 2>&1
RANGES=$(python $R/tools/pmc_top.py select $O/${TAG}_pmctrace "$RE" $NTOP $O/${TAG}_pmc_selection.json)
echo "selection: $(echo $RANGES | wc -w) launches"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "$RE" --kernel-iteration-range $RANGES --kernel-trace -d $O/${TAG}_pmc_$C --output-format csv -- python $R/tools/one_factorization.py --grid $GRID > $O/${TAG}_pmc_$C.log 2>&1
  echo "$C rc=$?"
done
# matrix-pipe utilisation of the same launches (north_star: "MFMA utilisation on fat ones"): a third pass
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-include-regex "$RE" --kernel-iteration-range $RANGES --kernel-trace -d $O/${TAG}_pmc_MFMA --output-format csv -- python $R/tools/one_factorization.py --grid $GRID > $O/${TAG}_pmc_MFMA.log 2>&1
echo "MFMA rc=$?"
cd $R
python tools/pmc_top.py summarise $O/${TAG}_pmc_selection.json $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE "poisson3d_${GRID}^3_geometricND_leaf4" $O/${TAG}_pmc_summary_poisson${GRID}_top${NTOP}.json $O/${TAG}_pmc_MFMA
