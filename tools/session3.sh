#!/bin/bash
# round-3 GPU session 3: where the h2 stack spends its time
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s3; mkdir -p $O; export TMPDIR=/tmp; cd $R
L=openmpl_amd/lib/libmpl_hip.so; cp $L build_tmp/lib_default.so
{
echo "== stack time by active teams (default build)"; timeout 300 python tools/stack_time.py
echo "== x3 for comparison"; ENGINE=x3 timeout 300 python tools/stack_time.py 4096 512
for v in abl1 abl2 abl3 abl4 abl8 abl16 nostag default; do
  cp build_tmp/lib_$v.so $L; echo "== variant $v"; timeout 200 python tools/stack_time.py 4096 512
done
cp build_tmp/lib_dbg.so $L; echo "== phases (dbg build)"; timeout 300 python tools/chain_phase.py 544 3 4096; timeout 300 python tools/chain_phase.py 544 3 512
cp build_tmp/lib_default.so $L
} > $O/time.log 2>&1
timeout 600 python -m pytest tests/test_h2_gpu.py -x -q > $O/h2.log 2>&1; tail -3 $O/h2.log
cat $O/time.log
