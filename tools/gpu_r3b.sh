# round-3 second GPU call: quadrilateral fit tests + timing, full-size parity re-run
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 500 python -m pytest tests/test_quad_gpu.py -x -q -m gpu -p no:cacheprovider --tb=short > $O/quad.log 2>&1; echo "quad rc=$?"; tail -25 $O/quad.log | cut -c1-300
step timeout -k 10 300 python -m pytest tests/test_fullsize_gpu.py tests/test_geometry_gpu.py -q -s -m gpu -p no:cacheprovider > $O/fullsize.log 2>&1; echo "fullsize rc=$?"; grep -E "^\[|passed|failed|Error|assert" $O/fullsize.log | cut -c1-230 | tail -40
step timeout -k 10 200 python tools/geom_aug_bench.py > $O/geom_aug_bench.log 2>&1; echo "geom rc=$?"; grep -v amdgpu.ids $O/geom_aug_bench.log | tail -14
