#!/bin/bash
# A/B of prebuilt library variants on the block-stack time (one gpurun call):  bash tools/ab_stack.sh [rounds] tag1 tag2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
L=openmpl_amd/lib/libmpl_hip.so; cp $L build_tmp/lib_default.so
N=$1; shift
for r in $(seq 1 $N); do
  for v in "$@"; do
    cp build_tmp/lib_$v.so $L
    echo "$v: $(timeout 200 python tools/stack_time.py 4096 512 2>/dev/null | tr '\n' ' ')"
  done
done
cp build_tmp/lib_default.so $L
