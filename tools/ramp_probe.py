"""Per-step time of the headline forward over a long run (HIP events per step): how long the clock governor takes to
reach the sustained state.  python tools/ramp_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, make_batch, model_flags
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
m = build_model(model_flags("chosen", 4, 12), torch.device("cuda"))
P, R, C = make_batch(1024, 4, "cuda", 1)
with torch.no_grad():
    m(P, rays=R, centers=C); torch.cuda.synchronize()
    time.sleep(2.0)                       # idle: let the part fall back to its idle state
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        m(P, rays=R, centers=C)
        ev[i + 1].record()
    torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print("first 16 steps (ms):", " ".join("%.3f" % x for x in t[:16]))
for lo, hi in ((0, 10), (10, 30), (30, 60), (60, 120), (120, 250), (250, 500), (500, 1000), (1000, N)):
    if lo < N:
        s = t[lo:min(hi, N)]
        print("steps %4d..%4d: %.4f ms per step (%.0f poses/s)" % (lo, min(hi, N), sum(s) / len(s), 1024e3 / (sum(s) / len(s))))
