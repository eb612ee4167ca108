R=$GRAFT_REPO_ROOT; cd $R
for W in 4 5 6; do echo "== LEAF_MINW $W"; CHOLMOD_HIP_LEAF_MINW=$W python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin" | head -2; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
