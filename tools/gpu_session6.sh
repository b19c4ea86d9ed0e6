#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/pytest6.log 2>&1; tail -3 gpurun_out/pytest6.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
python bench.py > gpurun_out/bench6.log 2>&1; tail -1 gpurun_out/bench6.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6 -o r01 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/prof6.log 2>&1
ls gpurun_out/prof6
P="rocprofv3 --kernel-trace --output-format csv"
$P --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc6a -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc6a.log 2>&1
$P --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc6b -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc6b.log 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/pmc6c -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc6c.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/pmc6d -o p -- python tools/gemm_ab.py 544 > gpurun_out/pmc6d.log 2>&1
ls gpurun_out/pmc6a gpurun_out/pmc6c
