# final library vs the library with the (invalid) buffer-store epilogue of the streaming kernels: what the fix costs on one box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f
rm -rf $O; mkdir -p $O
ALT=$R/vickers-hardness-unet_amd/libvkunet_alt.so
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
for i in 1 2 3; do
VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 > $O/bench_new_$i.log 2>&1; echo "final run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_new_$i.log)"
VK_LIB=$ALT VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 > $O/bench_old_$i.log 2>&1; echo "buffer-store run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_old_$i.log)"
done
