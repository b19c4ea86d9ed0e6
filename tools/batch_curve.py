"""Poses/s of the whole forward by batch size and views (CHOSEN flag set, depth 12).  python tools/batch_curve.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmpl_amd import cabi, detrng
from openmpl_amd.multiview_mpl import MultiView_MPL

lib = cabi.load()
dev = "cuda"
forms = [0]
for V in (2, 4):
    m = MultiView_MPL(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=12, num_views=V, pose_3d_emb_learnable=True)
    detrng.fill_module_(m, seed=11)
    m = m.to(dev).eval()
    for B in (1, 16, 32, 64, 128, 256, 384, 512, 640, 768, 1024, 1536, 2048, 4096, 8192):
        p, r, c = detrng.make_inputs(B, V, seed=1)
        P, R, C = ([torch.from_numpy(x).to(dev) for x in l] for l in (p, r, c))
        row = []
        for f in forms:
            with torch.no_grad():
                for _ in range(5): m(P, rays=R, centers=C)
                torch.cuda.synchronize()
                n = 30
                t0 = time.perf_counter()
                for _ in range(n): m(P, rays=R, centers=C)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
            row.append("%8.1f us %9.0f poses/s" % (dt * 1e6, B / dt))
        print("V=%d B=%4d | " % (V, B) + " | ".join(row))
