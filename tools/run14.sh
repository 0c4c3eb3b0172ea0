timeout -k 10 200 python tools/microbench.py --ops wgrad --only D3c1,D3c2,D4c1,D4c2 2>&1 | grep -v amdgpu.ids
