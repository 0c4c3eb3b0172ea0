mkdir -p gpurun_out
VK_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/bench_dist1.log | cut -c1-600
