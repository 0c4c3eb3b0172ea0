# every bench.py mode on the final code (one gpurun box):  gpurun --timeout 900 -- 'bash tools/gpu_modes.sh'
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --prof-steps 0"
echo "== train bf16 bs32";            timeout -k 10 200 $B --steps 20 --warmup 5 | cut -c1-200 &&
echo "== forced single-rank RCCL group" && VK_BENCH_FORCE_DIST=1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 $B --steps 20 --warmup 5 | cut -c1-200 &&
echo "== torchrun nproc 1" && timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 | cut -c1-200 &&
echo "== fp32 eval bs16" && timeout -k 10 200 $B --mode infer --dtype fp32 --batch 16 --steps 10 --warmup 3 | cut -c1-200 &&
echo "== bf16 eval bs16" && timeout -k 10 200 $B --mode infer --dtype bf16 --batch 16 --steps 20 --warmup 3 | cut -c1-200 &&
echo "== fp16 train 1024 bs8" && timeout -k 10 200 $B --dtype fp16 --size 1024 --batch 8 --steps 10 --warmup 3 | cut -c1-200
echo "== 2-rank rehearsal of the control flow (gloo, both ranks on cuda:0; not a measurement)" && VK_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --prof-steps 2 | cut -c1-260
echo "== 2-rank data-parallel invariants (gloo, both ranks on cuda:0)" && timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29566 tests/diag/dp_rehearsal.py 2>&1 | grep "identical"
