mkdir -p gpurun_out
for B in 256 512 1024; do
  VK_WH_BLOCKS=$B VK_PROF_DETAIL=1 timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_whb_$B.log 2>&1 || exit 1
done
