export TMPDIR=/tmp
timeout 300 python tools/chain_check.py 2>&1 | tail -6
timeout 300 python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['parity'])"
