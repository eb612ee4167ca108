# thin-regime check: launch profile of the G3_circuit stand-in + the parity tests that cover the thin kernels
R=$GRAFT_REPO_ROOT; cd $R
python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5
