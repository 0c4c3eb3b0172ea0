# streaming kernel at step level: bench A/B (VK_NO_STREAM=1 = tile kernels), then the full GPU suite
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3i
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
for v in "" 1 "" 1; do
  VK_NO_STREAM=$v
  if [ -z "$v" ]; then unset VK_NO_STREAM; else export VK_NO_STREAM; fi
  step timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --api-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; k = r['all_kernels_ms_per_step']
        sel = {n: v for n, v in k.items() if n.startswith(('stream', 'c16', 'col_16b_t16_bn16', 'col_16b_t16_bn32'))}
        print('VK_NO_STREAM=$v', d['ms_per_step'], 'ms/step', d['value'], 'img/s |', sel)
" | tee -a $O/bench_ab.log
done
unset VK_NO_STREAM
step timeout -k 10 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $O/tests_full.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_full.log | cut -c1-300
