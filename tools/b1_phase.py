"""Per-phase time of the bf16 block stack by stopping the persistent launch after n GEMMs (mpl_x3_stack_mode bits 8..):
    python tools/b1_phase.py [V] [B] [depth]
Prints the stack time for n = 1 .. 8 (two block applications): bf16 engine (default form), its pair form, the fp32 engine."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402

dev = torch.device("cuda", 0)
lib = cabi.load()
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2
fs = sys.argv[4] if len(sys.argv) > 4 else "chosen"
m = build_model(model_flags(fs, V, L), dev)
b = make_batch(B, V, dev, seed=1)
rows = []
for label, prec, bits in (("bf16", "bf16", 0), ("bf16 pairs", "bf16", 2 << 1), ("fp32", "fp32", 0)):
    m.set_matmul_precision(prec)
    ts = []
    for n in list(range(1, 9)) + [0]:
        cabi.check(lib.mpl_x3_stack_mode(bits | (n << 8)), "mode")
        with torch.no_grad():
            for i in range(3):
                m(b[0], rays=b[1], centers=b[2])
            torch.cuda.synchronize()
            cabi.profile_start()
            for i in range(6):
                m(b[0], rays=b[1], centers=b[2])
            torch.cuda.synchronize()
            pr = cabi.profile_stop()
        ts.append(pr["gemm"][0] / 6 * 1e3)
    cabi.check(lib.mpl_x3_stack_mode(0), "mode")
    d = [ts[0]] + [ts[i] - ts[i - 1] for i in range(1, 8)]
    print("%-8s V=%d B=%d L=%d  cumulative us: %s | whole %0.f" % (label, V, B, L, " ".join("%.0f" % t for t in ts[:8]), ts[8]))
    print("%-8s   per phase us (qkv proj fc1 fc2 | qkv proj fc1 fc2): %s" % ("", " ".join("%.0f" % t for t in d)), flush=True)
