cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/tools/gemm_one.py 188160 256 256 1 > $R/gpurun_out/pmcA.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcB -- python3 $R/tools/gemm_one.py 188160 256 256 1 > $R/gpurun_out/pmcB.log 2>&1
