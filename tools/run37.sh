mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x > gpurun_out/t37.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/t37.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
echo SIDE; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 2>&1 | grep '^{"metric"' | cut -c1-160
echo NOSIDE; VK_NO_SIDE_STREAM=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 2>&1 | grep '^{"metric"' | cut -c1-160
done
