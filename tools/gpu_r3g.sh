# (1) scalar-fma BN+ReLU transform in the pipelined K >= 128 kernel (VK_COL_DBG=32) in-process A/B; (2) bench after the BN finalize / coefficient load hoisting; (3) full GPU suite
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python tools/microbench.py --only L2,L3,L4,D0c1,D1c1 --ops fwd --ab VK_COL_DBG=0,32 --rounds 6 > $O/microbench_fma.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu.ids $O/microbench_fma.log | tail -8
for v in 0 32 0 32; do
  VK_COL_DBG=$v step timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --api-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; k = r['all_kernels_ms_per_step']
        print('VK_COL_DBG=$v', d['ms_per_step'], 'ms/step', d['value'], 'img/s | colq', k.get('colq_16b_t16_bn128_w8'), k.get('colq_16b_t16_bn128_w8_dgrad'), 'bn_finalize', k.get('bn_finalize'), 'bn_bwd_coeffs', k.get('bn_bwd_coeffs'))
" | tee -a $O/bench_ab.log
done
step timeout -k 10 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $O/tests_full.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests_full.log | cut -c1-300
