# same-box A/B of two libraries: libvkunet.so (tree) vs libvkunet_alt.so (the previous conv_halo.hip)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3x
rm -rf $O; mkdir -p $O
ALT=$R/vickers-hardness-unet_amd/libvkunet_alt.so
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
for i in 1 2; do
step timeout -k 10 300 python tools/microbench.py --only D4c1,D4c2,D3c2 --ops fwd,dgrad_bnr --reps 30 > $O/new$i.log 2>&1; echo "new rc=$?"; grep -v amdgpu.ids $O/new$i.log | tail -6 | cut -c1-60
VK_LIB=$ALT step timeout -k 10 300 python tools/microbench.py --only D4c1,D4c2,D3c2 --ops fwd,dgrad_bnr --reps 30 > $O/old$i.log 2>&1; echo "old rc=$?"; grep -v amdgpu.ids $O/old$i.log | tail -6 | cut -c1-60
done
for i in 1 2 3; do
VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_new_$i.log 2>&1; echo "new run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_new_$i.log)"
VK_LIB=$ALT VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_old_$i.log 2>&1; echo "old run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_old_$i.log)"
done
