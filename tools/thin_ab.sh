# A/B of thin-kernel builds: default against lib/libcholmod_amd_lds.so (-DTF_PANEL_LDS), per MINW.
R=$GRAFT_REPO_ROOT; cd $R
for LIB in "" "$R/suitesparse_amd/lib/libcholmod_amd_lds.so"; do
for W in 6 4; do echo "== lib=${LIB##*/} MINW $W"; CHOLMOD_AMD_LIB=$LIB CHOLMOD_HIP_THIN_MINW=$W python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "thin|launches"; done
done
echo "== parity with the lds lib"
CHOLMOD_AMD_LIB=$R/suitesparse_amd/lib/libcholmod_amd_lds.so python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
echo "== timing lds"; CHOLMOD_AMD_LIB=$R/suitesparse_amd/lib/libcholmod_amd_lds.so CHOLMOD_HIP_THIN_TIMING=1 python tools/launch_profile.py poisson2d 1259 8 2>&1 | grep -E "cycles"
