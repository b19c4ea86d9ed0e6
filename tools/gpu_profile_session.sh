#!/bin/bash
# One GPU session: headline bench, rocprofv3 kernel trace of the bench, PMC passes over the stand-alone GEMM shapes
# (split-operand kernels, the default fp32 path) and the SPT kernel.  Outputs under gpurun_out/ (session id 12);
# tools/make_profiles.py turns them into the tracked profiles/ files.
mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py > gpurun_out/bench12.log 2>&1; tail -1 gpurun_out/bench12.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof12 -o r01 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/prof12.log 2>&1
export MPL_GEMM_X3=1
P="rocprofv3 --kernel-trace --output-format csv"
$P --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc12a -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc12a.log 2>&1
$P --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc12b -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc12b.log 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/pmc12c -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc12c.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/pmc12d -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc12d.log 2>&1
unset MPL_GEMM_X3
$P --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d gpurun_out/pmc12e -o p -- python tools/spt_ab.py > gpurun_out/pmc12e.log 2>&1
