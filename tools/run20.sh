mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv_fwd or conv_dgrad" > gpurun_out/t20.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/t20.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/microbench.py --ops fwd,dgrad --reps 20 > gpurun_out/micro_col.log 2>&1 && cat gpurun_out/micro_col.log | grep -v amdgpu.ids
VK_COL_ALT=1 timeout -k 10 200 python tools/microbench.py --only L2,L3,D0c1,D1c1 --ops fwd,dgrad --reps 20 > gpurun_out/micro_col_alt1.log 2>&1 && echo ALT1 && cat gpurun_out/micro_col_alt1.log | grep -v amdgpu.ids
VK_COL_ALT=3 timeout -k 10 200 python tools/microbench.py --only L2,L3,L4,D0c1,D1c1 --ops fwd,dgrad --reps 20 > gpurun_out/micro_col_alt3.log 2>&1 && echo ALT3 && cat gpurun_out/micro_col_alt3.log | grep -v amdgpu.ids
VK_COL_ALT=2 timeout -k 10 200 python tools/microbench.py --only L4 --ops fwd,dgrad --reps 20 > gpurun_out/micro_col_alt2.log 2>&1 && echo ALT2 && cat gpurun_out/micro_col_alt2.log | grep -v amdgpu.ids
