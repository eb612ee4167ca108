# A/B of the persistent update kernel at the headline size: one bench line per variant
# usage (on the GPU box): bash tools/ab_persist.sh "<shape> <shape> ..."   (shape = tile:RxC:wpc:groups, "off" = flag 512)
cd $GRAFT_REPO_ROOT
for S in $1; do
  if [ "$S" = "off" ]; then F=""; unset CHOLMOD_HIP_PERSIST_SHAPE; else F="--hip-flags 512"; export CHOLMOD_HIP_PERSIST_SHAPE=$S; fi
  T=$(echo $S | tr ':' '_')
  python bench.py --no-cpu-baseline --steps 2 $F > gpurun_out/ab_$T.json 2> gpurun_out/ab_$T.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$T.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$S", "GF/s %.0f ms %.1f resid %.1e updTF %.2f" % (d["value"], d["ms_per_step"], d["residual_2norm"], r["achieved"]), {k: round(v,3) for k,v in r["seconds_by_class"].items() if k.startswith("update")})
except Exception as e:
    print("$S", "ERR", e)
PY
done
