#!/bin/bash
# A/B of prebuilt library variants on the whole forward (one gpurun call):  bash tools/ab_bench.sh rounds tag1 tag2 ...
# prints poses/s, ms per step and the per-kernel times of bench.py for every variant, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
L=openmpl_amd/lib/libmpl_hip.so; cp $L build_tmp/lib_default.so
N=$1; shift
for r in $(seq 1 $N); do
  for v in "$@"; do
    cp build_tmp/lib_$v.so $L
    timeout 300 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['parity']['max_scaled'])"
  done
done
cp build_tmp/lib_default.so $L
