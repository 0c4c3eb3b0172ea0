mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench1.log 2>&1
rc=$?; echo "bench rc=$rc"; tail -12 gpurun_out/bench1.log
