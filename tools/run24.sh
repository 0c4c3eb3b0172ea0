mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/tests_full.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/tests_full.log
