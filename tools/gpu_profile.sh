# bench.py default line + rocprofv3 kernel stats + PMC HBM-traffic passes on a gpurun box:  gpurun --timeout 1200 -- 'bash tools/gpu_profile.sh'
# then copy gpurun_out/prof3/runc/*_kernel_stats.csv, gpurun_out/traffic3.json and the bench line into profiles/
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( time timeout -k 10 500 python bench.py ) > gpurun_out/bench_default.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/bench_default.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof3 $R/gpurun_out/pmc_fetch3 $R/gpurun_out/pmc_write3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --prof-steps 0 > $R/gpurun_out/prof3.log 2>&1
echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 > $R/gpurun_out/pmc_fetch3.log 2>&1
echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 > $R/gpurun_out/pmc_write3.log 2>&1
echo "write rc=$?"
cd $R && python tools/pmc_traffic.py gpurun_out/pmc_fetch3 gpurun_out/pmc_write3 gpurun_out/traffic3.json | head -30
find gpurun_out/prof3 -name "*kernel_stats.csv" | head; du -sh gpurun_out/prof3 gpurun_out/pmc_fetch3 gpurun_out/pmc_write3
