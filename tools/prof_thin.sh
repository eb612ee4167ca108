cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for W in "poisson2d 1259 p2d" "poisson3d 100 p100"; do set -- $W
rocprofv3 --kernel-trace --stats -d $O/r02_stats_$3 --output-format csv -- python $R/tools/one_factorization.py --workload $1 --grid $2 --repeat 3 > $O/r02_stats_$3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r02_fetch_$3 --output-format csv -- python $R/tools/one_factorization.py --workload $1 --grid $2 --checks > $O/r02_fetch_$3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r02_write_$3 --output-format csv -- python $R/tools/one_factorization.py --workload $1 --grid $2 --checks > $O/r02_write_$3.log 2>&1
done
cd $R; python tools/pmc_by_kernel.py gpurun_out/r02_fetch_p2d gpurun_out/r02_write_p2d > gpurun_out/r02_pmc_p2d.json; python tools/pmc_by_kernel.py gpurun_out/r02_fetch_p100 gpurun_out/r02_write_p100 > gpurun_out/r02_pmc_p100.json; cat gpurun_out/r02_pmc_p2d.json; grep -h "factor_checks\|one factorization" gpurun_out/r02_fetch_*.log; ls gpurun_out/r02_stats_p2d/*/
