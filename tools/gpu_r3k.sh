# 64-channel streaming kernel: correctness, then in-process A/B against the tile kernels (VK_STREAM64=0)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider --tb=short -k "conv_fwd or conv_dgrad or fused or stream64" > $O/ops.log 2>&1; rc=$?; echo "ops rc=$rc"; tail -4 $O/ops.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^FAILED|^ERROR" $O/ops.log | head -30 | cut -c1-220; exit 1; fi
step timeout -k 10 300 python tools/microbench.py --only L1 --ops fwd,dgrad_bnr,dgrad_acc --ab VK_STREAM64=0,1 --rounds 5 > $O/microbench_ab.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu.ids $O/microbench_ab.log | tail -5
step timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short > $O/model.log 2>&1; echo "model rc=$?"; tail -4 $O/model.log | cut -c1-300
