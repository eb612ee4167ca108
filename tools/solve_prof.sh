R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r02j_stats_solve --output-format csv -- python $R/tools/solve_time.py 100 > $O/r02j_stats_solve.log 2>&1
tail -5 $O/r02j_stats_solve.log
f=$(ls $O/r02j_stats_solve/*/*kernel_stats.csv | tail -1); grep -E "solve|Name" $f | cut -c1-60,150-260
