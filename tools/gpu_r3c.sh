# staggered K >= 128 tile kernel: correctness (bit-identical to the pipelined kernel), per-layer A/B in one process, step A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short -k "staggered or stag" > $O/stag_tests.log 2>&1; rc=$?; echo "stag tests rc=$rc"; tail -12 $O/stag_tests.log | cut -c1-250
if [ $rc -ne 0 ]; then echo "correctness first: not timing a kernel that fails its tests"; exit 1; fi
step timeout -k 10 300 python tools/microbench.py --only L2,L3,L4,D0c1,D1c1 --ops fwd,dgrad,dgrad_bnr --ab VK_COL_PIPE=1,2 --rounds 5 > $O/microbench_ab.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu.ids $O/microbench_ab.log | tail -20
for v in 1 2 1 2; do
  VK_COL_PIPE=$v step timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --api-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('VK_COL_PIPE=$v', d['ms_per_step'], 'ms/step', d['value'], 'img/s | dominant', r['kernel'], r['avg_launch_ms'], 'ms', r['frac'], '|', {k: v for k, v in list(r['all_kernels_ms_per_step'].items())[:8]})
" | tee -a $O/bench_ab.log
done
