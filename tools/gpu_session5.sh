#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for v in 4 1; do MPL_GEMM_VAR=$v python -m pytest tests/test_gpu_parity.py -q -x -k "ln_linear or token_attention or forward_matches" 2>&1 | tail -3; done
for v in 1 4; do MPL_GEMM_VAR=$v python tools/gemm_ab.py 544; MPL_GEMM_VAR=$v python tools/gemm_ab.py 1088; done
MPL_GEMM_VAR=4 python tools/microbench.py
MPL_GEMM_VAR=4 MPL_ATT_V1=1 python tools/microbench.py | grep attention
} > gpurun_out/s5.log 2>&1
grep -v amdgpu.ids gpurun_out/s5.log
