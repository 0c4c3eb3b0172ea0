mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "wgrad" > gpurun_out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/tests.log
if [ $rc -eq 0 ]; then
  timeout -k 10 300 python tools/microbench.py --ops wgrad > gpurun_out/micro.log 2>&1
  echo rc=$?; grep -v amdgpu.ids gpurun_out/micro.log
fi
