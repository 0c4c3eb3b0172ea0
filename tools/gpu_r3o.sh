# slab reduces on the side stream (VK_SIDE_STREAM=2): model tests with the mode forced on, then same-box bench A/B (alternating)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
VK_SIDE_STREAM=2 step timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short > $O/model.log 2>&1; rc=$?; echo "model rc=$rc"; tail -3 $O/model.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2 3; do
for m in 0 2; do
VK_SIDE_STREAM=$m VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_m${m}_$i.log 2>&1; echo "mode $m run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_m${m}_$i.log)"
done
done
