R=$GRAFT_REPO_ROOT; cd $R
for W in 4,4,4 4,4,3 4,3,3 5,4,3 5,4,4 3,3,3; do echo "== MINW $W"; CHOLMOD_HIP_THIN_MINW=$W python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin"| head -6; done
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
python tools/bench_summary.py --workload poisson2d --grid 1259 --no-cpu-baseline --steps 10
python tools/bench_summary.py --grid 100 --no-cpu-baseline --steps 5
