mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/ops.log 2>&1
rc=$?
tail -5 gpurun_out/ops.log
if [ $rc -ne 124 ] && [ $rc -ne 137 ]; then
  timeout -k 10 200 python tools/debug_layers.py --n 2 --size 64 --dtype f32 > gpurun_out/dbg_eval_f32.log 2>&1
  echo "debug rc=$?"
  tail -3 gpurun_out/dbg_eval_f32.log
fi
