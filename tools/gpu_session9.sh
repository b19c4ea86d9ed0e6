#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
python tools/gemm_ab.py 544
MPL_GEMM_KG=9 python tools/gemm_ab.py 544
MPL_GEMM_KG=2 python tools/gemm_ab.py 544
python tools/gemm_ab.py 1088
MPL_GEMM_KG=9 python tools/gemm_ab.py 1088
python tools/gemm_ab.py 32
python tools/microbench.py | tail -9
} > gpurun_out/s9.log 2>&1
grep -v amdgpu.ids gpurun_out/s9.log
