# bench.py default line + rocprofv3 kernel stats + PMC HBM-traffic passes on a gpurun box:
#   rm -rf gpurun_out/profrun; gpurun --timeout 1200 -- 'bash tools/gpu_profile.sh'
# everything to keep lands in gpurun_out/profrun/summary/ (kernel_stats.csv, traffic.json, pmc csv, the bench lines): copy it into profiles/rNN/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profrun
rm -rf $O; mkdir -p $O/summary
( time timeout -k 10 500 python bench.py ) > $O/summary/bench_default.log 2>&1
echo "rc=$?"; tail -4 $O/summary/bench_default.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --prof-steps 0 > $O/stats.log 2>&1
echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 > $O/pmc_fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 > $O/pmc_write.log 2>&1
echo "write rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 --api-steps 0 > $O/pmc_sq.log 2>&1
echo "sq rc=$?"
cd $R && python tools/pmc_sq.py $O/pmc_sq > $O/summary/sq_counters.txt 2>&1; head -8 $O/summary/sq_counters.txt | cut -c1-300
cd $R && python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/summary/traffic.json | head -12
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/summary/kernel_stats.csv \;
# the bench line WITH the traffic of this very tree: install the fresh traffic file where bench.py looks for it and run the default bench again
ROUND=${VK_ROUND:-r03}; mkdir -p profiles/$ROUND; cp $O/summary/traffic.json profiles/$ROUND/traffic.json
( timeout -k 10 400 python bench.py ) > $O/summary/bench_with_traffic.log 2>&1
echo "rc=$?"; tail -1 $O/summary/bench_with_traffic.log | cut -c1-300
timeout -k 10 200 python tools/geom_aug_bench.py > $O/summary/geom_aug_bench.log 2>&1; tail -3 $O/summary/geom_aug_bench.log
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq      # raw traces stay on the box
