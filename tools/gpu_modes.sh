# every bench.py mode on the final code (one gpurun box):  gpurun --timeout 1100 -- 'bash tools/gpu_modes.sh'
# full JSON lines go to gpurun_out/modes_full.log, a short form to stdout
mkdir -p gpurun_out
OUT=gpurun_out/modes_full.log
: > $OUT
B="python bench.py --no-cpu-baseline --prof-steps 0"
run() { echo "== $1" | tee -a $OUT; shift; "$@" 2>/dev/null | tee -a $OUT | cut -c1-230; }
run "train bf16 bs32" timeout -k 10 200 $B --steps 20 --warmup 5 &&
run "forced single-rank RCCL group" env VK_BENCH_FORCE_DIST=1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 $B --steps 20 --warmup 5 --prof-steps 3 &&
run "torchrun nproc 1" timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 &&
run "fp32 eval bs16 (configs[1])" timeout -k 10 200 $B --mode infer --dtype fp32 --batch 16 --steps 10 --warmup 3 &&
run "fp32 eval bs1" timeout -k 10 200 $B --mode infer --dtype fp32 --batch 1 --steps 50 --warmup 5 &&
run "bf16 eval bs16" timeout -k 10 200 $B --mode infer --dtype bf16 --batch 16 --steps 20 --warmup 3 &&
run "fp16 train 1024 bs8 (configs[4] workload, loss scale 2^16)" timeout -k 10 200 $B --dtype fp16 --size 1024 --batch 8 --steps 10 --warmup 3
echo "== 2-rank rehearsal of the control flow (gloo, both ranks on cuda:0; not a measurement)" | tee -a $OUT
VK_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --prof-steps 2 --api-steps 2 2>/dev/null | tee -a $OUT | cut -c1-260
echo "== 2-rank data-parallel invariants (gloo, both ranks on cuda:0)" | tee -a $OUT
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29566 tests/diag/dp_rehearsal.py 2>&1 | grep -v "amdgpu.ids" | tee -a $OUT | tail -8
