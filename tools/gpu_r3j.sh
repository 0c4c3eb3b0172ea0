R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short > $O/model.log 2>&1; echo "model rc=$?"; tail -3 $O/model.log | cut -c1-300
for i in 1 2; do
step timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --api-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; k = r['all_kernels_ms_per_step']
        print(d['ms_per_step'], 'ms/step', d['value'], 'img/s | weights_repack', k.get('weights_repack'), 'sum', round(sum(k.values()), 3))
" | tee -a $O/bench.log
done
