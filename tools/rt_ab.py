"""One-tile vs two-tile stage of the block stack (h2_stack_kernel / h2_stack2_kernel), same process, alternating:
    python tools/rt_ab.py
Prints ms per forward and per stack launch for the shapes where teams own >= 2 row tiles."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402

dev = torch.device("cuda", 0)
lib = cabi.load()
CASES = [("full", 4, 12, 1024), ("chosen", 8, 12, 1024), ("chosen", 8, 2, 1024), ("chosen", 4, 12, 2048), ("chosen", 4, 12, 8192),
         ("full", 8, 2, 1024), ("chosen", 4, 12, 1536)]
for fs, V, L, B in CASES:
    m = build_model(model_flags(fs, V, L), dev)
    b = [make_batch(B, V, dev, seed=1, step=s) for s in range(2)]
    res = {}
    for rep in range(2):
        for rt in (1, 2):
            cabi.check(lib.mpl_x3_stack_mode(rt << 1), "mode")
            with torch.no_grad():
                for i in range(3):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                n = 10
                t0 = time.perf_counter()
                for i in range(n):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e3
                cabi.profile_start()
                for i in range(4):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                pr = cabi.profile_stop()
            res.setdefault(rt, []).append((dt, pr["gemm"][0] / 4))
    cabi.check(lib.mpl_x3_stack_mode(0), "mode")
    f = lambda rt: "%.3f ms / stack %.3f ms (%.0f poses/s)" % (min(x[0] for x in res[rt]), min(x[1] for x in res[rt]), B / min(x[0] for x in res[rt]) * 1e3)
    print("%-6s V=%d L=%2d B=%4d | one tile: %s | two tiles: %s" % (fs, V, L, B, f(1), f(2)), flush=True)
    del m
