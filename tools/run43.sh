mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv_fwd or conv_dgrad" > gpurun_out/t43.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/t43.log
[ $rc -eq 0 ] || exit $rc
VK_C16_TPW=3 timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv_fwd or conv_dgrad" > gpurun_out/t43b.log 2>&1
rc=$?; echo "tests(tpw=3) rc=$rc"; tail -3 gpurun_out/t43b.log
[ $rc -eq 0 ] || exit $rc
for t in 1 2 4 8; do echo "C16_TPW=$t"; VK_C16_TPW=$t timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 --only D4c1,D4c2 2>&1 | grep -v amdgpu.ids | grep -v "D4c1  fwd"; done
