VK_WH_MAXCOMBO=64 timeout -k 10 200 python tools/microbench.py --ops wgrad --only L4,D0c1 2>&1 | grep -v amdgpu.ids
