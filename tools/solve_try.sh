R=$GRAFT_REPO_ROOT; cd $R
python tools/solve_time.py 100
python tools/solve_time.py 160 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_and_demo.py tests/test_gpu_scale.py tests/test_complex.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin"
