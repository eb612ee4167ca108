#!/bin/bash
# round-5 measurement batch for the mid-size regime (GPU box; from the repo root): per-launch profiles of the three
# secondary workloads under the chain / update-kernel knobs, and a kernel-trace time line of the nd24k stand-in.
TAG=${1:-r05b}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for W in "box3d 42" "poisson3d 100" "poisson2d 1259"; do set -- $W
  for V in "" "CHOLMOD_HIP_CHAINF_AUTO=1" "CHOLMOD_HIP_UPD3_MIN_TILES=1024" "CHOLMOD_HIP_UPD3_MIN_TILES=512"; do
    echo "=== $1 $2 [$V]"
    env $V python tools/launch_profile.py $1 $2 999 2>/dev/null | head -24
  done
done > $O/${TAG}_launch_profiles.log 2>&1
cd /tmp && export TMPDIR=/tmp
for W in "box3d 42 box42r3" "poisson3d 100 p100" "poisson2d 1259 p2d1259"; do set -- $W
  rocprofv3 --kernel-trace -d $O/${TAG}_trace_$3 --output-format csv -- python $R/tools/one_factorization.py --workload $1 --grid $2 --repeat 3 > $O/${TAG}_trace_$3.log 2>&1
  python $R/tools/trace_gaps.py $O/${TAG}_trace_$3 > $O/${TAG}_gaps_$3.txt 2>&1
  rm -rf $O/${TAG}_trace_$3
done
cd $R
python - <<'PY' > $O/${TAG}_secondary_lines.json 2> $O/${TAG}_secondary.err
import json, bench
out = []
for wl, m in (("poisson3d", 100), ("box3d", 42), ("poisson2d", 1259)):
    out.append(bench.secondary_line(wl, m))
print(json.dumps(out))
PY
