#!/bin/bash
# Diagnostic builds of the library with the r03 streaming-kernel bug reconstructed (MUBUF stores in the epilogue) and one candidate
# remedy each; only conv_halo.hip is recompiled, the other objects are the product build's.  Output: $VK_DIAG_OUT/libvk_<name>.so
#   usage: bash tests/diag/build_stream_variants.sh            (container or GPU box; ~1 min per variant, 4 in parallel)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vickers-hardness-unet_amd/csrc
O=${VK_DIAG_OUT:-/tmp/vkdiag}
mkdir -p $O
(cd $C && make -s)
FLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
build() {   # name, extra defines
  /opt/rocm/bin/hipcc $FLAGS $2 -c $C/conv_halo.hip -o $O/conv_halo_$1.o
  OBJS=$(ls $C/*.o | grep -v conv_halo.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $O/conv_halo_$1.o -ldl -o $O/libvk_$1.so
  echo "built $O/libvk_$1.so"
}
# r03 = MUBUF stores, no scheduling barrier behind the row's MFMAs (VK_STREAM_DIAG bit 4): the shipped-then-fixed r03 code
build r03     "-DVK_STREAM_BUFSTORE -DVK_STREAM_DIAG=16" &      # expected: WRONG (rows y = 3 mod 4, element 3 of tile 1)
build r03_v0  "-DVK_STREAM_BUFSTORE -DVK_STREAM_DIAG=17" &      # + vmcnt(0) in front of every ring write: still wrong (not a load race)
build r03_l0  "-DVK_STREAM_BUFSTORE -DVK_STREAM_DIAG=18" &      # + lgkmcnt(0) behind every ring write: still wrong (not an LDS race)
build r03_s0  "-DVK_STREAM_BUFSTORE -DVK_STREAM_DIAG=20" &      # + vmcnt(0) behind every store: right (the asm statement moves the MFMA)
wait
build r03_px  "-DVK_STREAM_BUFSTORE -DVK_STREAM_DIAG=24" &      # stores under an exec predicate: right (different block structure)
build bufsb   "-DVK_STREAM_BUFSTORE" &                          # MUBUF stores + the r04 scheduling barrier: right (the fix, with the r03 stores)
wait
