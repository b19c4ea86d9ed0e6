"""Hand-off store policy of the persistent stack kernels, one process: plain stores for teams that sit on one XCD (default; decided in
the kernel from HW_REG_XCC_ID) against write-through always (mpl_x3_stack_mode bit 7).  Bitwise check + ms per stack launch.
    python tools/wt_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402

dev = torch.device("cuda", 0)
lib = cabi.load()
bad = 0
for fs, V, L, B, prec in [("chosen", 4, 12, 1024, "fp32"), ("full", 4, 12, 1024, "fp32"), ("chosen", 8, 12, 1024, "fp32"), ("chosen", 2, 12, 256, "fp32"),
                          ("chosen", 8, 2, 1024, "bf16"), ("chosen", 8, 12, 1024, "bf16"), ("chosen", 4, 12, 1024, "bf16"), ("chosen", 5, 2, 300, "fp32")]:
    m = build_model(model_flags(fs, V, L), dev)
    m.set_matmul_precision(prec)
    b = [make_batch(B, V, dev, seed=1, step=s) for s in range(2)]
    res, outs = {}, {}
    for rep in range(3):
        for tag, bits in (("plain", 0), ("wt", 1 << 7)):
            cabi.check(lib.mpl_x3_stack_mode(bits), "mode")
            with torch.no_grad():
                for i in range(3):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                outs[tag] = m(b[0][0], rays=b[0][1], centers=b[0][2]).clone()
                torch.cuda.synchronize()
                cabi.profile_start()
                for i in range(10):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                pr = cabi.profile_stop()
            res.setdefault(tag, []).append(pr["gemm"][0] / 10)
    cabi.check(lib.mpl_x3_stack_mode(0), "mode")
    same = torch.equal(outs["plain"], outs["wt"]) and bool(torch.isfinite(outs["plain"]).all())
    bad += not same
    print("%-6s V=%d L=%2d B=%4d %-5s | stack ms: plain %s | write-through %s | bitwise %s"
          % (fs, V, L, B, prec, " ".join("%.3f" % t for t in res["plain"]), " ".join("%.3f" % t for t in res["wt"]), same), flush=True)
    del m
print("failures:", bad)
sys.exit(1 if bad else 0)
