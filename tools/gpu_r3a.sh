# round-3 first GPU call: new parity tests, full suite, CU-hold experiment, SQ counter passes
#   gpurun --timeout 1200 -- 'bash tools/gpu_r3a.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
rm -rf $O; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping, no further GPU step"; exit $rc; fi; return $rc; }
step timeout -k 10 400 python -m pytest tests/test_fullsize_gpu.py -q -s -m gpu -p no:cacheprovider > $O/fullsize.log 2>&1; echo "fullsize rc=$?"; grep -E "^\[|passed|failed|Error|assert" $O/fullsize.log | cut -c1-260 | tail -60
step timeout -k 10 500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --deselect tests/test_fullsize_gpu.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
step timeout -k 10 120 python tests/diag/cu_hold.py > $O/cu_hold.log 2>&1; echo "cu_hold rc=$?"; grep -v amdgpu.ids $O/cu_hold.log | tail -12
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1; grep -c SQ_ $O/counters_list.txt
step timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 --api-steps 0 > $O/pmc_sq1.log 2>&1; echo "sq1 rc=$?"
step timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 --api-steps 0 > $O/pmc_sq2.log 2>&1; echo "sq2 rc=$?"; tail -3 $O/pmc_sq2.log | cut -c1-200
cd $R
python tools/pmc_sq.py $O/pmc_sq1 > $O/sq_counters_pass1.txt 2>&1; head -24 $O/sq_counters_pass1.txt | cut -c1-330
python tools/pmc_sq.py $O/pmc_sq2 > $O/sq_counters_pass2.txt 2>&1; head -12 $O/sq_counters_pass2.txt | cut -c1-330
rm -rf $O/pmc_sq1 $O/pmc_sq2
