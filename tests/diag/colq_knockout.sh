#!/bin/bash
# Knock-out timing of the pipelined K >= 128 loop (results of the diagnostic builds are WRONG by construction): which instruction class
# the ~600 cycles per stage beyond the 1,536 MFMA cycles belong to.  Builds: -DVK_COLQ_DIAG=1 (one weight DMA per wave and stage
# instead of three), 2 (no operand transform + halo store in the loop), 6 (also no halo loads), 7 (all three).
cd $GRAFT_REPO_ROOT
for L in vickers-hardness-unet_amd/libvkunet.so .diagbuild/libvk_colqdiag1.so .diagbuild/libvk_colqdiag2.so .diagbuild/libvk_colqdiag6.so .diagbuild/libvk_colqdiag7.so vickers-hardness-unet_amd/libvkunet.so; do
  echo "== $L"
  VK_LIB=$PWD/$L timeout -k 10 200 python tools/microbench.py --only L2,L3,D0c1 --ops fwd,dgrad --reps 30 2>&1 | grep -v amdgpu.ids || exit 1
done
