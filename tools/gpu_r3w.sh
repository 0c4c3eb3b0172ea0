R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3w
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider --tb=short -k "stream_conv or conv_fwd or conv_dgrad or fused" > $O/ops.log 2>&1; rc=$?; echo "ops rc=$rc"; tail -3 $O/ops.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $O/ops.log | head -40 | cut -c1-220; exit 1; fi
step timeout -k 10 600 python tools/microbench.py --only D4c1,D4c2,D3c2 --ops fwd,dgrad_bnr --reps 30 > $O/mb.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/mb.log | tail -6 | cut -c1-200
for i in 1 2 3; do
VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_$i.log 2>&1; echo "run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.log)"
done
