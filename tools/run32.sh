mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv_fwd or conv_dgrad" > gpurun_out/t32.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/t32.log
[ $rc -eq 0 ] || exit $rc
echo SPLIT; timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 --only L2,L3,D0c1,D1c1 2>&1 | grep -v amdgpu.ids
echo NOSPLIT; VK_COL_DBG=16 timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 --only L2,L3,D0c1,D1c1 2>&1 | grep -v amdgpu.ids
echo SPLIT_NOAFFINE; timeout -k 10 200 python tools/microbench.py --ops fwd --reps 20 --only L2,L3,D0c1,D1c1 --no-affine 2>&1 | grep -v amdgpu.ids
echo NOSPLIT_NOAFFINE; VK_COL_DBG=16 timeout -k 10 200 python tools/microbench.py --ops fwd --reps 20 --only L2,L3,D0c1,D1c1 --no-affine 2>&1 | grep -v amdgpu.ids
