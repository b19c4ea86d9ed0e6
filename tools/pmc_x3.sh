#!/bin/bash
# rocprofv3 counter passes over one x3 GEMM shape: bash tools/pmc_x3.sh proj
export TMPDIR=/tmp
W=${1:-proj}
R=$GRAFT_REPO_ROOT
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv"
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmcx3a -o p -- python $R/tools/x3_one.py $W > $R/gpurun_out/pmcx3a.log 2>&1
$P --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcx3b -o p -- python $R/tools/x3_one.py $W > $R/gpurun_out/pmcx3b.log 2>&1
$P --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES -d $R/gpurun_out/pmcx3c -o p -- python $R/tools/x3_one.py $W > $R/gpurun_out/pmcx3c.log 2>&1
