mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "wgrad" > gpurun_out/t48.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/t48.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/microbench.py --ops wgrad --reps 20 --only L1,L3,D3c1,D3c2,D4c1,D4c2 2>&1 | grep -v amdgpu.ids
