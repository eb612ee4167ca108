# run the world-4 native factorization until it comes back wrong, then print what every rank found
cd $GRAFT_REPO_ROOT
for i in $(seq 1 14); do
  rm -f /tmp/dd.*
  for r in 0 1 2 3; do
    RANK=$r WORLD_SIZE=4 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + i)) GLOO_SOCKET_IFNAME=lo DIST_TEST_EXCHANGE=native \
      CHOLMOD_HIP_RCCL_LIBRARY=$GRAFT_REPO_ROOT/tests/standin_rccl/libstandin_rccl.so CHOLMOD_HIP_UPD3_MIN_TILES=1 "$@" \
      timeout 120 python tools/dist_debug_worker.py /tmp/dd > /tmp/ddlog.$r 2>&1 &
  done
  wait
  if python - <<'PY'
import json, sys
bad = False
for r in range(4):
    try: d = json.load(open(f"/tmp/dd.{r}"))
    except Exception as e: print("rank", r, "no result", e); bad = True; continue
    if d["bad"] or d["status"] != 0: bad = True
sys.exit(1 if bad else 0)
PY
  then echo "run $i ok"; else echo "run $i WRONG"; cat /tmp/dd.0; echo; break; fi
done
