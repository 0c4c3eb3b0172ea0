R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short -k "staggered" > $O/stag_tests.log 2>&1; rc=$?; echo "stag tests rc=$rc"; tail -3 $O/stag_tests.log | cut -c1-250
if [ $rc -ne 0 ]; then exit 1; fi
step timeout -k 10 300 python tools/microbench.py --only L2,L3,L4,D0c1,D1c1 --ops fwd,dgrad --ab VK_COL_PIPE=1,2 --rounds 5 > $O/microbench_ab.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu.ids $O/microbench_ab.log | tail -12
VK_COL_DBG=16 step timeout -k 10 300 python tools/microbench.py --only L2,L3,L4,D0c1,D1c1 --ops fwd,dgrad --ab VK_COL_PIPE=1,2 --rounds 5 > $O/microbench_ab_prio.log 2>&1; echo "microbench prio rc=$?"; grep -v amdgpu.ids $O/microbench_ab_prio.log | tail -12
VK_LIB=$R/vickers-hardness-unet_amd/libvkunet_stamp.so step timeout -k 10 200 python tools/stamps_cols.py L2 L3 D0c1 > $O/stamps_cols.log 2>&1; echo "stamps rc=$?"; grep -v amdgpu.ids $O/stamps_cols.log | tail -8
VK_COL_DBG=16 VK_LIB=$R/vickers-hardness-unet_amd/libvkunet_stamp.so step timeout -k 10 200 python tools/stamps_cols.py L3 D0c1 > $O/stamps_cols_prio.log 2>&1; echo "stamps prio rc=$?"; grep -v amdgpu.ids $O/stamps_cols_prio.log | tail -8
