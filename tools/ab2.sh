# A/B of two prebuilt libraries inside ONE gpurun call (boxes differ by a few per cent):
#   bash tools/ab2.sh openmpl_amd/lib/A.so openmpl_amd/lib/B.so [rounds]
export TMPDIR=/tmp
L=openmpl_amd/lib/libmpl_hip.so
for r in $(seq 1 ${3:-2}); do
  for v in "$1" "$2"; do
    cp "$v" $L
    timeout 300 python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step']['gemm'], j['roofline']['kernel_ms_per_step']['spt'], j['parity']['max_scaled'])"
  done
done
