for d in 0 1 2 3 4 8 11; do echo "WH_DBG=$d"; VK_WH_DBG=$d timeout -k 10 100 python tools/microbench.py --ops wgrad --reps 20 --only L1,L3,D0c1 2>&1 | grep -v amdgpu.ids; done
echo NOEPI; VK_WH_DBG_NOEPI=1 timeout -k 10 100 python tools/microbench.py --ops wgrad --reps 20 --only L1,L3,D0c1 2>&1 | grep -v amdgpu.ids
