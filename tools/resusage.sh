#!/bin/bash
# usage: tools/resusage.sh <file.hip> [extra hipcc flags]  -- per-kernel registers / spills / scratch of one translation unit
f=$1; shift
cd "$(dirname "$0")/../openmpl_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$f" -o /tmp/resusage_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
python3 -c '
import sys, re, subprocess
cur = {}
rows = []
for line in sys.stdin:
    if "error" in line or "warning" in line: print(line.rstrip())
    m = re.search(r"remark:\s+(.*?):\s+(\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k] = v
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(.*", "", n)
    print("%-70s VGPR %4s AGPR %4s SGPR %4s spillS %4s spillV %4s scratch %4s LDS %s" % (n[-70:], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]")))
'
rm -f /tmp/resusage_$$.o
