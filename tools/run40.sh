mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv_fwd or conv_dgrad" > gpurun_out/t40.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/t40.log
[ $rc -eq 0 ] || exit $rc
VK_COL_TPW=3 timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv_fwd or conv_dgrad" > gpurun_out/t40b.log 2>&1
rc=$?; echo "tests(tpw=3) rc=$rc"; tail -3 gpurun_out/t40b.log
[ $rc -eq 0 ] || exit $rc
for t in 1 2 4 8; do echo "TPW=$t"; VK_COL_TPW=$t timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 --only D3c1,D3c2,D4c1 2>&1 | grep -v amdgpu.ids; done
echo "TPW=auto"; timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 --only D3c1,D3c2,D4c1 2>&1 | grep -v amdgpu.ids
