#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
MPL_GEMM_VAR=1 python tools/gemm_ab.py 544
python tools/gemm_ab.py 544
MPL_GEMM_NG=1 python tools/gemm_ab.py 544
MPL_GEMM_NG=2 python tools/gemm_ab.py 544
python tools/gemm_ab.py 1088
python tools/gemm_ab.py 544 1024
python tools/gemm_ab.py 544 8192
python tools/microbench.py
python tools/microbench.py --flagset full | tail -8
} > gpurun_out/s7.log 2>&1
grep -v amdgpu.ids gpurun_out/s7.log
