"""Per-kernel breakdown of the joints x views grid (KPTOK) forward at V=31, B=256 (BASELINE configs[4], literal form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, make_batch, model_flags, timed_steps
from openmpl_amd import cabi
V, B = int(sys.argv[1]) if len(sys.argv) > 1 else 31, int(sys.argv[2]) if len(sys.argv) > 2 else 256
f = model_flags("chosen", V, 12, FPT_blocks_view_keypoint_tokens=True)
m = build_model(f, torch.device("cuda"))
b = [make_batch(B, V, "cuda", seed=3000, step=s) for s in range(2)]
print("poses/s", B * 5 / timed_steps(m, b, 5, 2))
cabi.profile_start()
with torch.no_grad():
    for i in range(3): m(b[0][0], rays=b[0][1], centers=b[0][2])
torch.cuda.synchronize()
print({k: (round(t / 3, 3), n // 3) for k, (t, n) in cabi.profile_stop().items()})
