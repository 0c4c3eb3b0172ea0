for d in 0 1 2 4 6 8 14; do echo "DBG=$d"; VK_COL_DBG=$d timeout -k 10 100 python tools/microbench.py --ops fwd --reps 20 --only L3,D0c1 2>&1 | grep -v amdgpu.ids; done
