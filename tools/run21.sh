mkdir -p gpurun_out
for L in L1 L2 L3 D0c1; do VK_LIB=$GRAFT_REPO_ROOT/vickers-hardness-unet_amd/libvkunet_stamp.so timeout -k 10 120 python tools/stamps.py $L 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_col.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench_col.log | cut -c1-1500
timeout -k 10 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > gpurun_out/tests_full.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/tests_full.log
