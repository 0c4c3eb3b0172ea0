mkdir -p gpurun_out
timeout -k 10 300 python bench.py --mode infer --dtype fp32 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_infer_f32.log 2>&1
echo "rc=$?"; grep '^{"metric"' gpurun_out/bench_infer_f32.log | cut -c1-1200
timeout -k 10 300 python bench.py --mode infer --dtype bf16 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_infer_bf16.log 2>&1
echo "rc=$?"; grep '^{"metric"' gpurun_out/bench_infer_bf16.log | cut -c1-400
timeout -k 10 300 python bench.py --size 1024 --batch 8 --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_1024_f16.log 2>&1
echo "rc=$?"; grep '^{"metric"' gpurun_out/bench_1024_f16.log | cut -c1-700
VK_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 > gpurun_out/bench_dist1.log 2>&1
echo "rc=$?"; grep '^{"metric"' gpurun_out/bench_dist1.log | cut -c1-300
