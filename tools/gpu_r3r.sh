# streaming weight-gradient kernel for the 32-channel-chunk decoder layers: tests, then in-process A/B against the tile kernels
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3r
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider --tb=short -k "wgrad" > $O/ops.log 2>&1; rc=$?; echo "ops rc=$rc"; tail -3 $O/ops.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $O/ops.log | head -40 | cut -c1-220; exit 1; fi
step timeout -k 10 600 python tools/microbench.py --only D3c2,D4c1 --ops wgrad --ab VK_NO_WSTREAM=1, --rounds 5 > $O/microbench_ab.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu.ids $O/microbench_ab.log | tail -4
step timeout -k 10 600 python tools/microbench.py --only D3c2 --ops wgrad --ab VK_WS_KW=16,32 --rounds 5 > $O/microbench_kw.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu.ids $O/microbench_kw.log | tail -4
