for L in L1 L3 D0c1; do VK_LIB=$GRAFT_REPO_ROOT/vickers-hardness-unet_amd/libvkunet_stamp.so timeout -k 10 120 python tools/stamps.py $L 2>&1 | grep -v amdgpu.ids; done
