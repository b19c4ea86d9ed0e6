"""SPT time by batch: the fp32-MFMA kernel with its sequences spread over the chip (spt_kernel, p.spw sequences per workgroup)
against the packed-operand kernel (spt3_kernel, always 16).  MPL_SPT_SPREAD_MAX=0 keeps the packed kernel for "fp32".
    python tools/spt_spread.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402
dev = torch.device("cuda", 0)
for fs, V in (("chosen", 4), ("chosen", 2), ("full", 4)):
    m = build_model(model_flags(fs, V, 12), dev)
    for B in (1, 8, 16, 64, 96, 128, 192, 256, 384, 512, 1024):
        b = make_batch(B, V, dev, seed=B, step=0)
        row = []
        for prec in ("fp32", "fp32_mfma"):
            m.set_matmul_precision(prec)
            with torch.no_grad():
                for i in range(3):
                    m(b[0], rays=b[1], centers=b[2])
                torch.cuda.synchronize()
                cabi.profile_start()
                for i in range(8):
                    m(b[0], rays=b[1], centers=b[2])
                torch.cuda.synchronize()
                pr = cabi.profile_stop()
            row.append("%s SPT %.0f us" % (prec, pr["spt"][0] / 8 * 1e3))
        print("%-6s V=%d B=%4d (%5d sequences) | " % (fs, V, B, B * V) + " | ".join(row), flush=True)
    del m
