# how often a world-4 factorization over shared GPU 0 comes back wrong: variants by environment
cd $GRAFT_REPO_ROOT
run() { # name, count, env...
  local name=$1; local cnt=$2; shift; shift
  local fails=0
  for i in $(seq 1 $cnt); do
    env "$@" timeout 120 python - <<'PY' >/dev/null 2>&1 || fails=$((fails+1))
import sys, os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_dist as T
env = dict(T.NATIVE if os.environ.get("NATIVE_X", "1") == "1" else {}, CHOLMOD_HIP_UPD3_MIN_TILES=os.environ.get("MINT", "1"))
if os.environ.get("FLAGS"): env["CHOLMOD_TEST_HIP_FLAGS"] = os.environ["FLAGS"]
res = T._run_ranks(int(os.environ.get("WORLD", "4")), "gpu", "p3d_32", extra_env=env)
ok = all(r["ok"] == 1 and r["status"] == 0 and r["err"] < 1e-12 for r in res)
sys.exit(0 if ok else 1)
PY
  done
  echo "$name: $fails failures of $cnt"
}
"$@"
