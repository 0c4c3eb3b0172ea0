# weight gradient: where the fixed cost per launch goes (epilogue / slab / reduce), in-process A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python tools/microbench.py --only L1,L3,D0c1 --ops wgrad --ab VK_WH_DBG_NOEPI=,1 --rounds 3 > $O/noepi.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/noepi.log | tail -3
step timeout -k 10 300 python tools/microbench.py --only L1,L3,D0c1 --ops wgrad --ab VK_WH_NO_SLAB=,1 --rounds 3 > $O/noslab.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/noslab.log | tail -3
step timeout -k 10 300 python tools/microbench.py --only L1,L3,D0c1 --ops wgrad --ab VK_WH_BLOCKS=256,128,64 --rounds 3 > $O/blocks.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/blocks.log | tail -3
step timeout -k 10 300 python tools/microbench.py --only L1,L3,D0c1 --ops wgrad --prof --reps 10 > $O/prof.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/prof.log | tail -12
