R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_edge_and_demo.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
for F in 0 1024; do
python tools/bench_summary.py --workload box3d --grid 42 --no-cpu-baseline --steps 10 --hip-flags $F
python tools/bench_summary.py --grid 100 --no-cpu-baseline --steps 5 --hip-flags $F
python tools/bench_summary.py --workload poisson2d --grid 1259 --no-cpu-baseline --steps 10 --hip-flags $F
done
python tools/launch_profile.py box3d 42 2>&1 | head -12
