mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/tests_full.log 2>&1
echo "tests rc=$?"; grep -v "^  File\|^Extension" gpurun_out/tests_full.log | tail -25 | cut -c1-250
