mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "config2 or config3" > gpurun_out/tests_full.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/tests_full.log
