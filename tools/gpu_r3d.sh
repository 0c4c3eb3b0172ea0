# stamps of the staggered kernel + CU-hold experiment with the reducer policies
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
VK_LIB=$R/vickers-hardness-unet_amd/libvkunet_stamp.so step timeout -k 10 200 python tools/stamps_cols.py L2 L3 L4 D0c1 > $O/stamps_cols.log 2>&1; echo "stamps rc=$?"; grep -v amdgpu.ids $O/stamps_cols.log | tail -12
step timeout -k 10 300 python tests/diag/cu_hold.py > $O/cu_hold.log 2>&1; echo "cu_hold rc=$?"; grep -v amdgpu.ids $O/cu_hold.log | tail -24
