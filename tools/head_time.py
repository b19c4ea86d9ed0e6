"""Quick timing line of one configuration for A/B runs (tools/ab.sh): median ms per forward over regions of 20 + per-kernel split.
   [V=4 B=1024 DEPTH=12 FLAGSET=chosen PREC=fp32] python tools/head_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, make_batch, model_flags, timed_regions  # noqa: E402
from openmpl_amd import cabi  # noqa: E402
V, B, L = int(os.environ.get("V", 4)), int(os.environ.get("B", 1024)), int(os.environ.get("DEPTH", 12))
fs, prec = os.environ.get("FLAGSET", "chosen"), os.environ.get("PREC", "fp32")
dev = torch.device("cuda", 0)
m = build_model(model_flags(fs, V, L), dev)
m.set_matmul_precision(prec)
b = [make_batch(B, V, dev, seed=1000, step=s) for s in range(4)]
t = timed_regions(m, b, B, 20, 10, int(os.environ.get("REGIONS", 9)))
with torch.no_grad():
    cabi.profile_start()
    for i in range(40):
        m(b[i % 4][0], rays=b[i % 4][1], centers=b[i % 4][2])
    torch.cuda.synchronize()
    pr = cabi.profile_stop()
print("%s V=%d B=%d depth %d %s: median %.4f ms (min %.4f max %.4f) = %.1f k poses/s | kernels us: %s" % (
    fs, V, B, L, prec, t["median_ms"], t["min_ms"], t["max_ms"], t["poses_per_s"] / 1e3,
    " ".join("%s %.1f" % (k, v[0] / 40 * 1e3) for k, v in pr.items() if v[1])), flush=True)
