#!/bin/bash
# Knock-out timing of the pipelined K >= 128 loop (results of the diagnostic builds are WRONG by construction): which instruction class
# the ~600 cycles per stage beyond the 1,536 MFMA cycles belong to.  Builds (conv_halo.hip only, the other objects are the product
# build's; the r03 DMA placement, -DVK_COLQ_ILV=0, is the form the switches live in): -DVK_COLQ_DIAG=1 (one weight DMA per wave and
# stage instead of three), 2 (no operand transform + halo store in the loop), 6 (also no halo loads), 7 (all three).
#   usage (container, then GPU box):  bash tests/diag/colq_knockout.sh build  &&  gpurun -- 'bash tests/diag/colq_knockout.sh'
# Results of r04: profiles/r04/colq_knockout.log (and colq_bytes_knockout.log for the same instructions without their L2 traffic).
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vickers-hardness-unet_amd/csrc
O=$R/.diagbuild
if [ "$1" = "build" ]; then
  mkdir -p $O
  (cd $C && make -s)
  FLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
  OBJS=$(ls $C/*.o | grep -v conv_halo.o)
  for v in 0 1 2 6 7; do
    ( /opt/rocm/bin/hipcc $FLAGS -DVK_COLQ_ILV=0 -DVK_COLQ_DIAG=$v -c $C/conv_halo.hip -o /tmp/conv_halo_d$v.o &&
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/conv_halo_d$v.o -ldl -o $O/libvk_colqdiag$v.so && echo "built $O/libvk_colqdiag$v.so" ) &
  done
  wait
  exit 0
fi
cd $R
for v in 0 1 2 6 7 0; do
  echo "== VK_COLQ_DIAG=$v"
  VK_LIB=$O/libvk_colqdiag$v.so timeout -k 10 200 python tools/microbench.py --only L2,L3,D0c1 --ops fwd,dgrad --reps 30 2>&1 | grep -v amdgpu.ids || exit 1
done
