# strip height of the streaming convolution kernels (VK_STREAM_RS): in-process sweep, then bench A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3u
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python tools/microbench.py --only D4c1,D4c2,D3c2 --ops fwd,dgrad_bnr --ab VK_STREAM_RS=32,64,128,256,512 --rounds 3 > $O/rs.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/rs.log | tail -6 | cut -c1-330
for i in 1 2; do
for m in 32 128 256; do
VK_STREAM_RS=$m VK_BENCH_SKIP_CPU=1 step timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_rs${m}_$i.log 2>&1; echo "RS=$m run $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_rs${m}_$i.log)"
done
done
