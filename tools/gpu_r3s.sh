R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s
rm -rf $O; mkdir -p $O
ALT=$R/vickers-hardness-unet_amd/libvkunet_alt.so
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
for i in 1 2; do
step timeout -k 10 300 python tools/microbench.py --only D3c2,D4c1 --ops wgrad --reps 30 > $O/base$i.log 2>&1; echo "base rc=$?"; grep -v amdgpu.ids $O/base$i.log | tail -2
VK_LIB=$ALT step timeout -k 10 300 python tools/microbench.py --only D3c2,D4c1 --ops wgrad --reps 30 > $O/alt$i.log 2>&1; echo "alt(3 waves/SIMD) rc=$?"; grep -v amdgpu.ids $O/alt$i.log | tail -2
done
VK_LIB=$ALT step timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider --tb=short -k "wgrad_stream" > $O/ops_alt.log 2>&1; echo "alt ops rc=$?"; tail -2 $O/ops_alt.log | cut -c1-200
