# eval hipGraph replay: test, then batch-1 / batch-2 inference latency with and without it (same box), and batch 16 (graph not used)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short -k "graph or eval_logits or head" > $O/test.log 2>&1; rc=$?; echo "test rc=$rc"; tail -5 $O/test.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E " $O/test.log | head -20 | cut -c1-200; exit 1; fi
step timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short -k "head" > $O/head.log 2>&1; echo "head ops rc=$?"; tail -2 $O/head.log | cut -c1-200
for dt in bf16 fp32; do
for b in 1 2; do
for g in 0 1; do
if [ $g = 0 ]; then export VK_NO_GRAPH=1; else unset VK_NO_GRAPH; fi
step timeout -k 10 300 python bench.py --mode infer --batch $b --dtype $dt --steps 200 --warmup 20 > $O/infer_${dt}_b${b}_g$g.log 2>&1; echo "$dt batch $b graph $g rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/infer_${dt}_b${b}_g$g.log)"
done
done
done
unset VK_NO_GRAPH
step timeout -k 10 300 python bench.py --mode infer --batch 16 --dtype bf16 --steps 50 --warmup 10 > $O/infer_bf16_b16.log 2>&1; echo "bf16 batch 16 rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/infer_bf16_b16.log)"
