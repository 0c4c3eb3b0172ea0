# batched weight gradients: whole backward in one batch vs one batch per stage vs one launch per layer (same box)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3z
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short > $O/model.log 2>&1; rc=$?; echo "model rc=$rc"; tail -3 $O/model.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $O/model.log | head -30 | cut -c1-220; exit 1; fi
step timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider --tb=short -k "wgrad_batch" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/ops.log | cut -c1-200
for i in 1 2 3; do
for m in layer stage all; do
unset VK_NO_WGRAD_BATCH VK_BACKWARD_PER_STAGE
if [ $m = layer ]; then export VK_NO_WGRAD_BATCH=1; fi
if [ $m = stage ]; then export VK_BACKWARD_PER_STAGE=1; fi
VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_${m}_$i.log 2>&1; echo "$m run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${m}_$i.log) $(grep -o '"api_path_ms_per_step": [0-9.]*' $O/bench_${m}_$i.log)"
done
done
