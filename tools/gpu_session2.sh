#!/bin/bash
# GPU session 2: variant A/B of the GEMM, bench line, rocprofv3 kernel trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for v in 0 1 2 3; do MPL_GEMM_VAR=$v python tools/gemm_ab.py 544; done
for v in 0 1; do for a in 1 2; do MPL_GEMM_VAR=$v MPL_GEMM_ABL=$a python tools/gemm_ab.py 544; done; done
MPL_GEMM_VAR=1 python tools/gemm_ab.py 1088
MPL_GEMM_VAR=0 python tools/gemm_ab.py 1088
} > gpurun_out/gemm_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/gemm_ab.log
python tools/microbench.py > gpurun_out/micro2.log 2>&1; grep -v amdgpu.ids gpurun_out/micro2.log
python bench.py --steps 30 --warmup 5 > gpurun_out/bench2.log 2>&1; tail -2 gpurun_out/bench2.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o r01 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/prof_r01.log 2>&1
ls -R gpurun_out/prof_r01 | head -30
