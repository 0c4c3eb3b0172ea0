mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "head or loss or forward_backward or trajectory or eval_logits" > gpurun_out/t47.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/t47.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_step']; print(d['value'], d['ms_per_step'], {x:k[x] for x in k if 'head' in x or 'maxpool' in x or 'stem' in x or 'loss' in x})"; done
