# pixel-split weight-gradient variant (VK_WH_PS=1): correctness with the switch forced on, then in-process A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
export VK_WH_PS=1
step timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider --tb=short -k "wgrad" > $O/ops.log 2>&1; rc=$?; echo "ops rc=$rc"; tail -3 $O/ops.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^FAILED|^ERROR" $O/ops.log | head -30 | cut -c1-220; exit 1; fi
unset VK_WH_PS
step timeout -k 10 600 python tools/microbench.py --only L1,L2,L3,L4,D0c1,D1c1,D2c1 --ops wgrad --ab VK_WH_PS=0,1 --rounds 5 > $O/microbench_ab.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu.ids $O/microbench_ab.log | tail -8
