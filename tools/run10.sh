mkdir -p gpurun_out
timeout -k 10 300 python tools/microbench.py > gpurun_out/micro.log 2>&1
echo rc=$?; cat gpurun_out/micro.log | grep -v amdgpu.ids
