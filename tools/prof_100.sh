# Poisson 100^3 (configs[1]) by kernel, and the HBM traffic of its extend-add / update kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r02m_stats_p100 --output-format csv -- python $R/tools/one_factorization.py --workload poisson3d --grid 100 --repeat 3 > $O/r02m_stats_p100.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r02m_fetch_p100 --output-format csv -- python $R/tools/one_factorization.py --workload poisson3d --grid 100 --repeat 1 > $O/r02m_fetch_p100.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r02m_write_p100 --output-format csv -- python $R/tools/one_factorization.py --workload poisson3d --grid 100 --repeat 1 > $O/r02m_write_p100.log 2>&1
cd $R
python tools/pmc_by_kernel.py --second-half gpurun_out/r02m_fetch_p100 gpurun_out/r02m_write_p100 > gpurun_out/r02m_pmc_by_kernel_poisson100.json
f=$(ls $O/r02m_stats_p100/*/*kernel_stats.csv | tail -1); head -10 $f | cut -c1-70,120-220
grep -A4 "k_extend_add\|k_update2<64\|k_update2f\|k_trsm_upd" gpurun_out/r02m_pmc_by_kernel_poisson100.json | head -40
