echo NEW_NOBNR; VK_NO_BNR_FUSION=1 timeout -k 10 120 python tools/grad_err.py 2>&1 | grep -v amdgpu.ids | tail -4
echo NEW_ALT3; VK_COL_ALT=3 timeout -k 10 120 python tools/grad_err.py 2>&1 | grep -v amdgpu.ids | tail -4
echo NEW_ALT2; VK_COL_ALT=2 timeout -k 10 120 python tools/grad_err.py 2>&1 | grep -v amdgpu.ids | tail -4
