set -x
mkdir -p gpurun_out
export VK_LIB=$PWD/vickers-hardness-unet_amd/libvkunet_stamp.so
for L in L1 D2c1 L2; do
VK_COL_PERSIST=0 VK_COL_PIPE=0 timeout -k 10 120 python tools/stamps.py $L
done
unset VK_LIB
timeout -k 10 200 python tools/microbench.py --only L1,L2,D2c1 --ops fwd,dgrad --reps 30
timeout -k 10 200 python tools/microbench.py --only L1 --ops fwd --reps 30 --ab VK_COL_PERSIST=1,0
timeout -k 10 200 python tools/microbench.py --only L1 --ops fwd --reps 30 --no-stats
timeout -k 10 200 python tools/microbench.py --only L1 --ops fwd --reps 30 --no-affine --no-stats
