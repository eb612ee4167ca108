R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_edge_and_demo.py tests/test_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
for F in 0 4096; do echo "== flags $F"; HIP_FLAGS=$F python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin" | head -4; done
python tools/bench_summary.py --workload poisson2d --grid 1259 --no-cpu-baseline --steps 10
python tools/bench_summary.py --workload poisson2d --grid 1259 --no-cpu-baseline --steps 10 --hip-flags 4096
