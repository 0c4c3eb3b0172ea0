# software-pipelined fragment reads in the weight-gradient tile kernel: correctness, then A/B against the per-step build (libvkunet_b.so)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short -k "wgrad or comm" > $O/wgrad_tests.log 2>&1; rc=$?; echo "wgrad tests rc=$rc"; tail -5 $O/wgrad_tests.log | cut -c1-250
if [ $rc -ne 0 ]; then exit 1; fi
for rep in 1 2; do
for lib in libvkunet.so libvkunet_b.so; do
  echo "== $lib (rep $rep)" | tee -a $O/microbench_wgrad.log
  VK_LIB=$R/vickers-hardness-unet_amd/$lib step timeout -k 10 200 python tools/microbench.py --only L1,L2,L3,L4,D0c1,D1c1,D2c1 --ops wgrad --reps 30 2>/dev/null | tee -a $O/microbench_wgrad.log | cut -c1-150
done; done
for lib in libvkunet.so libvkunet_b.so libvkunet.so libvkunet_b.so; do
  VK_LIB=$R/vickers-hardness-unet_amd/$lib step timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --api-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; k = r['all_kernels_ms_per_step']
        print('$lib', d['ms_per_step'], 'ms/step', d['value'], 'img/s | wgrad_halo_16b_64x64ts', k.get('wgrad_halo_16b_64x64ts'), 'slab_reduce', k.get('wgrad_slab_reduce'), '32x64ts', k.get('wgrad_halo_16b_32x64ts'))
" | tee -a $O/bench_ab.log
done
