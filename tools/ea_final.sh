R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5
python tools/bench_summary.py --workload box3d --grid 42 --no-cpu-baseline --steps 10 | cut -c1-200
python tools/bench_summary.py --grid 100 --no-cpu-baseline --steps 5 | cut -c1-200
python tools/bench_summary.py --workload poisson2d --grid 1259 --no-cpu-baseline --steps 10 | cut -c1-200
python tools/launch_profile.py box3d 42 2>&1 | grep -E "extend_add n=|sum ms"
python tools/launch_profile.py poisson3d 100 2>&1 | grep -E "extend_add n=|sum ms"
python tools/launch_profile.py poisson2d 1259 2>&1 | grep -E "extend_add n=|sum ms"
