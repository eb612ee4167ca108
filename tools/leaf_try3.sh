R=$GRAFT_REPO_ROOT; cd $R
for W in 2 3 4; do echo "== LEAF_MINW $W"; CHOLMOD_HIP_LEAF_MINW=$W python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin" | head -2; done
for W in 3,3,3 4,4,3; do echo "== THIN_MINW $W"; CHOLMOD_HIP_THIN_MINW=$W python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin" | head -6; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
