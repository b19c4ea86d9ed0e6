# bench-only: A/B of build-time variants of the x3 engine on the headline workload (GEMM ms per forward)
export TMPDIR=/tmp
for f in "$@"; do
  export MPL_HIPCC_FLAGS="$f"   # exported: the measuring process hashes the flags too (build.source_hash)
  python -m openmpl_amd.build --force > /dev/null 2>&1
  timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step']['gemm'], j['parity']['max_scaled'])"
done
