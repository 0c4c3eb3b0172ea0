# data-parallel path (single-rank RCCL group): stage groups of the reducer policy vs one call per stage; 2-rank gloo invariants
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4b
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
export VK_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in 1 2 3; do
for m in stage group; do
unset VK_BACKWARD_PER_STAGE
if [ $m = stage ]; then export VK_BACKWARD_PER_STAGE=1; fi
MASTER_PORT=$((29600 + i * 2 + ${#m})) step timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 --prof-steps 0 > $O/bench_${m}_$i.log 2>&1; echo "$m run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${m}_$i.log)"
done
done
unset VK_BENCH_FORCE_DIST RANK WORLD_SIZE LOCAL_RANK VK_BACKWARD_PER_STAGE
step timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29566 tests/diag/dp_rehearsal.py > $O/rehearsal.log 2>&1; echo "rehearsal rc=$?"; grep -v "amdgpu.ids" $O/rehearsal.log | tail -2
