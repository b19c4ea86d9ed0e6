#!/bin/bash
# correctness + time of a prebuilt variant in one call:  bash tools/ab_check.sh tag [tag ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
L=openmpl_amd/lib/libmpl_hip.so; cp $L build_tmp/lib_default.so
for v in "$@"; do
  cp build_tmp/lib_$v.so $L
  echo "== $v: $(timeout 200 python tools/stack_time.py 4096 512 2>/dev/null | tr '\n' ' ')"
  timeout 600 python -m pytest tests/test_h2_gpu.py tests/test_gpu_parity.py -x -q -k "h2 or golden or launch_modes or independent" 2>&1 | tail -2
done
cp build_tmp/lib_default.so $L
