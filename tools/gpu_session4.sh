#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for a in 0 1 9 13 29 25 17 5 2; do MPL_GEMM_VAR=1 MPL_GEMM_ABL=$a python tools/gemm_ab.py 544; done
} > gpurun_out/gemm_abl.log 2>&1
grep -v amdgpu.ids gpurun_out/gemm_abl.log
