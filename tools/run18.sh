mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/prof2
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/microbench.py --only L1,L3 --ops fwd,dgrad,wgrad --reps 3 > $R/gpurun_out/pmc_fetch.log 2>&1
echo rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/microbench.py --only L1,L3 --ops fwd,dgrad,wgrad --reps 3 > $R/gpurun_out/pmc_write.log 2>&1
echo rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --prof-steps 0 > $R/gpurun_out/prof2.log 2>&1
echo rc=$?
