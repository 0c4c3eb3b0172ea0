mkdir -p gpurun_out
( time timeout -k 10 500 python bench.py ) > gpurun_out/bench_default.log 2>&1
echo "rc=$?"; grep -E "^\[bench|real" gpurun_out/bench_default.log | tail -8
timeout -k 10 300 python bench.py --mode infer --dtype fp32 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_infer_f32.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/bench_infer_f32.log | cut -c1-900
timeout -k 10 300 python bench.py --mode infer --dtype bf16 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_infer_bf16.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/bench_infer_bf16.log | cut -c1-300
