# fused eval decoder tail: test, then eval throughput with / without it (same box)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short -k "tail_fusion or eval_logits" > $O/test.log 2>&1; rc=$?; echo "test rc=$rc"; tail -4 $O/test.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  " $O/test.log | head -20 | cut -c1-220; exit 1; fi
for i in 1 2; do
for b in 16 1; do
for g in 1 0; do
if [ $g = 1 ]; then export VK_NO_TAIL_FUSION=1; else unset VK_NO_TAIL_FUSION; fi
step timeout -k 10 300 python bench.py --mode infer --batch $b --dtype bf16 --steps 100 --warmup 10 --no-cpu-baseline > $O/infer_b${b}_nofuse${g}_$i.log 2>&1; echo "bf16 batch $b no_fusion=$g run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/infer_b${b}_nofuse${g}_$i.log)"
done
done
done
