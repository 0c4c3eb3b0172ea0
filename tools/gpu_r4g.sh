R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4g
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
for i in 1 2; do
for m in 1 4 12 32; do
VK_WB_MINSEG=$m step timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 > $O/bench_ms${m}_$i.log 2>&1; echo "MINSEG=$m run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ms${m}_$i.log) $(grep -o '"wgrad_halo_16b_64x64ts_batch": [0-9.]*' $O/bench_ms${m}_$i.log)"
done
done
