R=$GRAFT_REPO_ROOT
for tag in A B; do echo "== old library $tag (A: before the branch-free epilogue, B: before the strip-height change)"; VK_LIB=$R/vickers-hardness-unet_amd/libvkunet_old$tag.so timeout -k 10 300 python tests/diag/stream_determinism_diag.py 2>&1 | grep -v amdgpu.ids | grep "dec3_conv2" ; done
