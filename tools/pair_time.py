"""ms per stack launch of the shapes whose teams walk pairs of row tiles (fp16x2 engine), for tools/ab.sh runs:  python tools/pair_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402

dev = torch.device("cuda", 0)
out = []
for fs, V, L, B in [("full", 4, 12, 1024), ("chosen", 8, 12, 1024), ("chosen", 4, 12, 2048), ("chosen", 4, 12, 1024)]:
    m = build_model(model_flags(fs, V, L), dev)
    b = make_batch(B, V, dev, seed=1)
    with torch.no_grad():
        m.set_matmul_precision("fp32_mfma")
        ref = m(b[0], rays=b[1], centers=b[2]).double()
        m.set_matmul_precision("fp32")
        for i in range(3):
            o = m(b[0], rays=b[1], centers=b[2])
        torch.cuda.synchronize()
        cabi.profile_start()
        for i in range(8):
            m(b[0], rays=b[1], centers=b[2])
        torch.cuda.synchronize()
        pr = cabi.profile_stop()
    err = float((o.double() - ref).abs().max() / ref.abs().max())
    out.append("%s V=%d B=%d: %.3f ms (err %.0e)" % (fs, V, B, pr["gemm"][0] / 8, err))
    del m
print(" | ".join(out))
