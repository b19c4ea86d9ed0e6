"""Per-kernel time of one forward for a (views, depth, batch, precision, flag set) other than the headline:
    python tools/cfg_breakdown.py 8 2 1024 bf16 [chosen|full|kptok]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, make_batch, model_flags
from openmpl_amd import cabi
V, L, B, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
kind = sys.argv[5] if len(sys.argv) > 5 else "chosen"
more = dict(FPT_blocks_view_keypoint_tokens=True) if kind == "kptok" else {}
m = build_model(model_flags("chosen" if kind == "kptok" else kind, V, L, **more), torch.device("cuda"))
m.set_matmul_precision(prec)
P, R, C = make_batch(B, V, "cuda", 1)
with torch.no_grad():
    for _ in range(5): m(P, rays=R, centers=C)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m(P, rays=R, centers=C)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    cabi.profile_start()
    for _ in range(5): m(P, rays=R, centers=C)
    torch.cuda.synchronize()
    prof = cabi.profile_stop()
print("V=%d depth=%d B=%d %s %s: %.3f ms per forward = %.0f poses/s ; per-kernel ms: %s" %
      (V, L, B, prec, kind, dt * 1e3, B / dt, {k: (round(t / 5, 4), n // 5) for k, (t, n) in prof.items()}))
