#!/bin/bash
# Two-tile stage experiments (one gpurun call): stack time at M = 8192 (pairs) and 4096 (single tiles) per prebuilt variant,
# then the per-wave-role stage breakdown of the debug builds.   bash tools/ab_rt2.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
L=openmpl_amd/lib/libmpl_hip.so; cp $L build_tmp/lib_default.so
for r in 1 2; do
  for v in default abl8 abl2 abl5 abl32; do
    cp build_tmp/lib_$v.so $L
    echo "$v: $(timeout 200 python tools/stack_time.py 8192 4096 2>/dev/null | tr '\n' ' ')"
  done
done
for v in dbg; do
  cp build_tmp/lib_$v.so $L
  echo "== $v, M = 8192 (two-tile stage)"; timeout 200 python tools/chain_phase.py 544 3 8192 2>&1 | tail -15
done
cp build_tmp/lib_dbg.so $L
echo "== dbg, M = 4096 (one-tile stage)"; timeout 200 python tools/chain_phase.py 544 3 4096 2>&1 | tail -15
cp build_tmp/lib_default.so $L
