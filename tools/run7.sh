mkdir -p gpurun_out
VK_PROF_DETAIL=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_detail.log 2>&1
echo "rc=$?"
