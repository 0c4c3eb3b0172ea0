mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv_fwd or conv_dgrad" > gpurun_out/t30.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/t30.log
[ $rc -eq 0 ] || exit $rc
echo SPREAD; timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 --only L1,L2,L3,L4,D0c1,D1c1 2>&1 | grep -v amdgpu.ids
echo BURST; VK_COL_BURST=1 timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 --only L1,L2,L3,L4,D0c1,D1c1 2>&1 | grep -v amdgpu.ids
for L in L3 D0c1; do VK_LIB=$GRAFT_REPO_ROOT/vickers-hardness-unet_amd/libvkunet_stamp.so timeout -k 10 120 python tools/stamps.py $L 2>&1 | grep -v amdgpu.ids; done
