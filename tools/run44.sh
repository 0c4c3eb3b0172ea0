mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "bn_add_relu or eval_logits or forward_backward or trajectory" > gpurun_out/t44.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/t44.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'bn_add_relu', d['roofline']['all_kernels_ms_per_step'].get('bn_add_relu'))"; done
