# HBM traffic of k_update2 at the headline size (Poisson 200^3): counters on the 48
# longest launches of one factorization only (74 % of the kernel's time; a PMC pass over
# all 3090 launches does not finish: 2 x 9 minutes cut off, as in round 1) --
# rocprofv3 --kernel-iteration-range, indices taken from a --kernel-trace of the same run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
timeout 540 rocprofv3 --pmc $C --kernel-include-regex k_update2 --kernel-iteration-range "[95-95]" "[141-141]" "[145-145]" "[205-205]" "[207-207]" "[209-209]" "[218-218]" "[308-308]" "[372-372]" "[429-429]" "[431-431]" "[495-495]" "[558-558]" "[559-559]" "[605-605]" "[669-669]" "[733-733]" "[797-797]" "[861-861]" "[925-925]" "[962-962]" "[1056-1056]" "[1164-1164]" "[1169-1169]" "[1178-1178]" "[1267-1267]" "[1331-1331]" "[1389-1389]" "[1453-1453]" "[1517-1517]" "[1561-1561]" "[1657-1657]" "[1766-1766]" "[1769-1769]" "[1779-1779]" "[1870-1870]" "[1934-1934]" "[1991-1991]" "[2055-2055]" "[2141-2141]" "[2205-2205]" "[2269-2269]" "[2333-2333]" "[2397-2397]" "[2461-2461]" "[2525-2525]" "[2589-2589]" "[2653-2653]" --kernel-trace -d $O/r02_pmc200_$C --output-format csv -- python $R/tools/one_factorization.py --grid 200 > $O/r02_pmc200_$C.log 2>&1
echo "$C rc=$?"; tail -2 $O/r02_pmc200_$C.log
done
cd $R; python tools/pmc_by_kernel.py gpurun_out/r02_pmc200_FETCH_SIZE gpurun_out/r02_pmc200_WRITE_SIZE
