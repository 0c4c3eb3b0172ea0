mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc1 $R/gpurun_out/pmc2
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/microbench.py --only ${1:-L3} --ops ${2:-fwd} --reps 3 > $R/gpurun_out/pmc1.log 2>&1
echo rc=$?
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/microbench.py --only ${1:-L3} --ops ${2:-fwd} --reps 3 > $R/gpurun_out/pmc2.log 2>&1
echo rc=$?
