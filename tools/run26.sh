mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/tests_full.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/tests_full.log
timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 --only L1,L2,L3,L4,D0c1,D1c1,D2c1 > gpurun_out/micro_col2.log 2>&1 && cat gpurun_out/micro_col2.log | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_col2.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_col2.log | cut -c1-1800
