"""Forward time by batch size: persistent block stack (mpl_x3_stack_mode 0) against one launch per GEMM (1).
    python tools/mode_by_batch.py [V]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402
dev = torch.device("cuda", 0)
lib = cabi.load()
V = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = build_model(model_flags("chosen", V, 12), dev)
for B in (64, 128, 256, 512, 1024, 2048):
    b = make_batch(B, V, dev, seed=B, step=0)
    row = []
    for mode in (0, 1, 0, 1):
        cabi.check(lib.mpl_x3_stack_mode(mode), "mode")
        with torch.no_grad():
            for i in range(5):
                m(b[0], rays=b[1], centers=b[2])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(50):
                m(b[0], rays=b[1], centers=b[2])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 50
        row.append("%s %.3f ms" % ("stack" if mode == 0 else "per-GEMM", dt * 1e3))
    cabi.check(lib.mpl_x3_stack_mode(0), "mode")
    print("V=%d B=%4d (%5d rows): " % (V, B, B * V) + " | ".join(row), flush=True)
