#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
python tools/gemm_ab.py 544
MPL_GEMM_KG=1 python tools/gemm_ab.py 544
MPL_GEMM_KG=2 python tools/gemm_ab.py 544
python tools/gemm_ab.py 1088
MPL_GEMM_KG=1 python tools/gemm_ab.py 1088
for a in 1 2 3; do MPL_GEMM_ABL=$a python tools/gemm_ab.py 544; done
python tools/microbench.py | tail -9
} > gpurun_out/s8.log 2>&1
grep -v amdgpu.ids gpurun_out/s8.log
