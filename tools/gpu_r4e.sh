R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python tests/diag/stream_determinism_diag.py > $O/det.log 2>&1; grep -v amdgpu.ids $O/det.log | grep -c "differing from run 0: \[\]"; grep -v amdgpu.ids $O/det.log | grep -v "differing from run 0: \[\]" | head
step timeout -k 10 300 python tests/diag/eval_determinism_diag.py > $O/evaldet.log 2>&1; grep -v amdgpu.ids $O/evaldet.log | tail -18
step timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -x > $O/all.log 2>&1; rc=$?; echo "all rc=$rc"; tail -3 $O/all.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $O/all.log | head -30 | cut -c1-220; exit 1; fi
