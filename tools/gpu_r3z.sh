# batched weight gradients in the step: model tests, then same-box bench A/B (VK_NO_WGRAD_BATCH=1 = one launch per layer)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3z
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short > $O/model.log 2>&1; rc=$?; echo "model rc=$rc"; tail -3 $O/model.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $O/model.log | head -30 | cut -c1-220; exit 1; fi
for i in 1 2 3; do
for m in 1 0; do
if [ $m = 1 ]; then export VK_NO_WGRAD_BATCH=1; else unset VK_NO_WGRAD_BATCH; fi
VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_nobatch${m}_$i.log 2>&1; echo "no_batch=$m run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_nobatch${m}_$i.log)"
done
done
