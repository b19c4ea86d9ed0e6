#!/bin/bash
# Package power and clocks (rocm-smi) beside a forward that loops back to back, + its ms per forward.
#   bash tools/power_probe.sh [views batch depth flagset precision seconds]      (on the GPU box; with tools/ab.sh for library variants)
cd "$(dirname "$0")/.."
V=${1:-4}; B=${2:-1024}; L=${3:-12}; FS=${4:-chosen}; PREC=${5:-fp32}; T=${6:-16}
python - $V $B $L $FS $PREC $T <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
from bench import build_model, make_batch, model_flags
V, B, L, FS, PREC, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], float(sys.argv[6])
dev = torch.device("cuda", 0)
m = build_model(model_flags(FS, V, L), dev)
m.set_matmul_precision(PREC)
b = make_batch(B, V, dev, seed=1, step=0)
n = 0
with torch.no_grad():
    for _ in range(20):
        m(b[0], rays=b[1], centers=b[2])
    torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < T:
        for _ in range(100):
            m(b[0], rays=b[1], centers=b[2])
        torch.cuda.synchronize()
        n += 100
    dt = time.time() - t0
print("loop: V=%d B=%d depth %d %s %s: %.4f ms per forward over %.0f s" % (V, B, L, FS, PREC, dt / n * 1e3, dt), flush=True)
PY
PID=$!
sleep 9
for i in 1 2 3; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|Package Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '
  echo
  sleep 1.5
done
wait $PID
