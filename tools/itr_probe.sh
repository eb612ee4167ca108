cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
i=0
for V in "[60-71]" "60-71" "[60,71]" "[1-5]"; do i=$((i+1))
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_update2 --kernel-iteration-range "$V" --kernel-trace -d $O/itr_$i --output-format csv -- python $R/tools/one_factorization.py --workload poisson2d --grid 1259 > $O/itr_$i.log 2>&1
echo "variant $V rc=$? files: $(ls $O/itr_$i/*/ 2>/dev/null | tr '\n' ' ')"; f=$(ls $O/itr_$i/*/*counter_collection.csv 2>/dev/null); [ -n "$f" ] && wc -l $f
done
