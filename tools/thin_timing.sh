R=$GRAFT_REPO_ROOT; cd $R
CHOLMOD_HIP_THIN_TIMING=1 python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin|cycles"
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5
