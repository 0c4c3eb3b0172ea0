echo DEFAULT; timeout -k 10 200 python tools/microbench.py --ops dgrad --reps 20 --only L1,D2c1,D3c1,D3c2 2>&1 | grep -v amdgpu.ids
echo ALT3; VK_COL_ALT=3 timeout -k 10 200 python tools/microbench.py --ops dgrad --reps 20 --only D2c1,D3c1 2>&1 | grep -v amdgpu.ids
echo ALT1; VK_COL_ALT=1 timeout -k 10 200 python tools/microbench.py --ops dgrad --reps 20 --only D2c1,D3c1 2>&1 | grep -v amdgpu.ids
