# weight gradient, deep-pipelined variant (libvkunet_alt.so built with -DVK_WH_DEEP=1): correctness, then same-box comparison
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
rm -rf $O; mkdir -p $O
ALT=$R/vickers-hardness-unet_amd/libvkunet_alt.so
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
VK_LIB=$ALT step timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider --tb=short -k "wgrad" > $O/ops.log 2>&1; rc=$?; echo "ops rc=$rc"; tail -3 $O/ops.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^FAILED|^ERROR" $O/ops.log | head -30 | cut -c1-220; exit 1; fi
for i in 1 2; do
step timeout -k 10 300 python tools/microbench.py --only L1,L2,L3,L4,D0c1,D1c1 --ops wgrad --reps 30 > $O/base$i.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/base$i.log | tail -6
VK_LIB=$ALT step timeout -k 10 300 python tools/microbench.py --only L1,L2,L3,L4,D0c1,D1c1 --ops wgrad --reps 30 > $O/alt$i.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/alt$i.log | tail -6
done
