R=$GRAFT_REPO_ROOT; cd $R
for V in ea8 ea4; do
  LIB=""; [ -n "$V" ] && LIB=$R/suitesparse_amd/lib/libcholmod_amd_$V.so
  echo "== EA variant ${V:-16}"
  CHOLMOD_AMD_LIB=$LIB python tools/launch_profile.py box3d 42 2>&1 | grep -E "extend_add n=|sum ms"
  CHOLMOD_AMD_LIB=$LIB python tools/launch_profile.py poisson3d 100 2>&1 | grep -E "extend_add n=|sum ms"
  CHOLMOD_AMD_LIB=$LIB python tools/launch_profile.py poisson2d 1259 2>&1 | grep -E "extend_add n=|sum ms"
done
