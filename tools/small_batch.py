"""Latency of small batches: the small-batch engine (sm_stack.hip) against the team kernels (mpl_x3_stack_mode bit 3), per-call
synchronised.   python tools/small_batch.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402
dev = torch.device("cuda", 0)
lib = cabi.load()
for fs, V, Bs in (("chosen", 2, (1, 8, 16, 32)), ("chosen", 4, (1, 4, 8, 12, 16)), ("full", 4, (1, 4, 8)), ("chosen", 8, (1, 2, 8))):
    m = build_model(model_flags(fs, V, 12), dev)
    for B in Bs:
        b = [make_batch(B, V, dev, seed=B, step=s) for s in range(2)]
        res = {}
        for mode, tag in ((0, "small-batch engine"), (8, "team kernels")):
            cabi.check(lib.mpl_x3_stack_mode(mode), "mode")
            with torch.no_grad():
                for i in range(5):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(40):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                    torch.cuda.synchronize()
                lat = (time.perf_counter() - t0) / 40 * 1e6
                t0 = time.perf_counter()
                for i in range(40):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                thr = (time.perf_counter() - t0) / 40 * 1e6
                cabi.profile_start()
                for i in range(4):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                pr = cabi.profile_stop()
            res[tag] = (lat, thr, pr["gemm"][0] / 4 * 1e3, pr["spt"][0] / 4 * 1e3)
        cabi.check(lib.mpl_x3_stack_mode(0), "mode")
        # the registered torch operator (openmpl_amd::forward) against the direct ctypes call: host time per call, not synchronised
        host = {}
        for route in (False, True, "auto"):
            m.use_torch_op(route)
            with torch.no_grad():
                for i in range(5):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(200):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                host[route] = (time.perf_counter() - t0) / 200 * 1e6
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(100):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                    torch.cuda.synchronize()
                host[(route, "sync")] = (time.perf_counter() - t0) / 100 * 1e6
        m.use_torch_op("auto")
        res["host us per call direct / via torch op"] = (host[False], host[True], host["auto"], host[(False, "sync")], host[(True, "sync")], host[("auto", "sync")])
        print("%-6s V=%d B=%2d | " % (fs, V, B) + " | ".join("%s: %.0f us per call (%.0f back to back; stack %.0f, SPT %.0f)" % ((t,) + res[t]) for t in res if not t.startswith("host")) +
              " | host enqueue per call: ctypes %.1f us, Python op openmpl_amd::forward %.1f us, C++ op openmpl_amd::lift (default) %.1f us; synchronised per call: %.0f / %.0f / %.0f us" % res["host us per call direct / via torch op"], flush=True)
    del m
