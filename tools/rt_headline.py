"""Headline shape (V=4, B=1024, CHOSEN, depth 12): one-tile stage on 256 workgroups vs two-tile stage on 128 (forced), alternating.
    python tools/rt_headline.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402
dev = torch.device("cuda", 0)
lib = cabi.load()
for fs, V, L, B in (("chosen", 4, 12, 1024), ("chosen", 2, 12, 1024), ("chosen", 4, 12, 512), ("chosen", 4, 12, 768), ("full", 4, 12, 512)):
    m = build_model(model_flags(fs, V, L), dev)
    b = [make_batch(B, V, dev, seed=1, step=s) for s in range(4)]
    for rep in range(3):
        for rt in (1, 2):
            cabi.check(lib.mpl_x3_stack_mode(rt << 1), "mode")
            with torch.no_grad():
                for i in range(10):
                    m(b[i % 4][0], rays=b[i % 4][1], centers=b[i % 4][2])
                torch.cuda.synchronize()
                n = 50
                t0 = time.perf_counter()
                for i in range(n):
                    m(b[i % 4][0], rays=b[i % 4][1], centers=b[i % 4][2])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e3
                cabi.profile_start()
                for i in range(4):
                    m(b[i % 4][0], rays=b[i % 4][1], centers=b[i % 4][2])
                torch.cuda.synchronize()
                pr = cabi.profile_stop()
            print("%s V=%d B=%d rt=%d: %.4f ms per forward (%.0f poses/s), stack %.4f ms" % (fs, V, B, rt, dt, B / dt * 1e3, pr["gemm"][0] / 4), flush=True)
    cabi.check(lib.mpl_x3_stack_mode(0), "mode")
    del m
