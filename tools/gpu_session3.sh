#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak > gpurun_out/mfma_peak.log 2>&1; cat gpurun_out/mfma_peak.log
rocprofv3 -L > gpurun_out/counters.txt 2>&1
grep -c "" gpurun_out/counters.txt
P="rocprofv3 --kernel-trace --output-format csv"
MPL_GEMM_VAR=1 $P --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc1 -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc1.log 2>&1
MPL_GEMM_VAR=1 $P --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU -d gpurun_out/pmc2 -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc2.log 2>&1
MPL_GEMM_VAR=1 $P --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d gpurun_out/pmc3 -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc3.log 2>&1
tail -3 gpurun_out/pmc1.log gpurun_out/pmc2.log gpurun_out/pmc3.log | cut -c1-200
find gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 -type f | head
