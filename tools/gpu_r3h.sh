# streaming small-channel conv kernel: correctness on the op tests + model tests, then in-process A/B against the tile kernels (VK_NO_STREAM)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider --tb=short -k "conv_fwd or conv_dgrad or upsample or pool2 or fused" > $O/ops.log 2>&1; rc=$?; echo "ops rc=$rc"; tail -15 $O/ops.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^FAILED|^ERROR" $O/ops.log | head -30 | cut -c1-200; exit 1; fi
step timeout -k 10 300 python tools/microbench.py --only D3c2,D4c1,D4c2 --ops fwd,dgrad,dgrad_bnr --ab VK_NO_STREAM=,1 --rounds 5 > $O/microbench_ab.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu.ids $O/microbench_ab.log | tail -12
step timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short > $O/model.log 2>&1; echo "model rc=$?"; tail -4 $O/model.log | cut -c1-300
