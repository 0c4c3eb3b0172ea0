echo NEW; timeout -k 10 120 python tools/grad_err.py 2>&1 | grep -v amdgpu.ids
echo OLD; VK_HALO_ROWSTAGED=1 timeout -k 10 120 python tools/grad_err.py 2>&1 | grep -v amdgpu.ids
echo TAP; VK_NO_HALO=1 timeout -k 10 120 python tools/grad_err.py 2>&1 | grep -v amdgpu.ids
