# Thin-regime check on the GPU box: parity tests that cover the thin kernels, then launch profiles.
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out
python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_edge_and_demo.py tests/test_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for W in 6 5 4; do echo "== MINW $W"; CHOLMOD_HIP_THIN_MINW=$W python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin|launches"; done
echo "== timing"; CHOLMOD_HIP_THIN_TIMING=1 python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin|cycles"
