mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x > gpurun_out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/tests.log
if [ $rc -eq 0 ]; then
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench2.log 2>&1
  echo "bench rc=$?"; tail -2 gpurun_out/bench2.log
fi
