"""Row-narrow teams of the fp16x2 block stack (h2_stackn_kernel): bitwise check against the whole-tile form + timing, one process:
    python tools/narrow_check.py
mpl_x3_stack_mode bits 5, 6: 1 = whole tiles always, 2 / 3 = 32- / 16-row workgroups where legal; bit 3: no small-batch engine; bit 4: the 16-row teams in the ring form (h2n_gemm.hip) instead of the direct-W form (h2d_gemm.hip)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402

dev = torch.device("cuda", 0)
lib = cabi.load()
bad = 0
CASES = [("chosen", 2, 12, 256), ("chosen", 4, 12, 256), ("chosen", 2, 12, 32), ("chosen", 8, 2, 64), ("chosen", 4, 2, 100),
                    ("full", 4, 2, 64), ("chosen", 2, 12, 1024), ("chosen", 4, 12, 512), ("chosen", 2, 2, 7), ("chosen", 16, 2, 24)]
if "--one" in sys.argv:          # timing runs of library variants (tools/ab.sh): the shipped call shape only
    CASES = CASES[:1]
for fs, V, L, B in CASES:
    m = build_model(model_flags(fs, V, L), dev)
    b = [make_batch(B, V, dev, seed=1, step=s) for s in range(2)]
    res, outs = {}, {}
    for rep in range(2):
        for tag, bits in (("whole", 1 << 5), ("rows32", 2 << 5), ("rows16", 3 << 5), ("rows16ring", (3 << 5) | 16), ("auto", 0)):
            cabi.check(lib.mpl_x3_stack_mode(bits | 8), "mode")
            with torch.no_grad():
                for i in range(3):
                    o = m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                outs[tag] = m(b[0][0], rays=b[0][1], centers=b[0][2]).clone()
                torch.cuda.synchronize()
                n = 20
                t0 = time.perf_counter()
                for i in range(n):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e3
                cabi.profile_start()
                for i in range(6):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                pr = cabi.profile_stop()
            res.setdefault(tag, []).append((dt, pr["gemm"][0] / 6))
    cabi.check(lib.mpl_x3_stack_mode(0), "mode")
    same = all(torch.equal(outs["whole"], outs[t]) for t in ("rows32", "rows16", "rows16ring", "auto")) and bool(torch.isfinite(outs["whole"]).all())
    bad += not same
    f = lambda k: "%.3f/%.3f ms %6.0f k/s" % (min(x[0] for x in res[k]), min(x[1] for x in res[k]), B / min(x[0] for x in res[k]))
    print("%-6s V=%2d L=%2d B=%4d | whole %s | 32 rows %s | 16 rows %s | 16 rows ring form %s | auto %s | bitwise %s" % (fs, V, L, B, f("whole"), f("rows32"), f("rows16"), f("rows16ring"), f("auto"), same),
          flush=True)
    del m
print("failures:", bad)
sys.exit(1 if bad else 0)
