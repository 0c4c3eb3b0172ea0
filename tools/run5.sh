mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --prof-steps 0 > $R/gpurun_out/prof1.log 2>&1
echo "prof rc=$?"
cd $R; find gpurun_out/prof1 -type f | head; tail -2 gpurun_out/prof1.log
