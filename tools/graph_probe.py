"""Does replaying the forward as a captured HIP graph beat stream launches?  (inter-kernel gaps on one stream)
    python tools/graph_probe.py [B]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import detrng  # noqa: E402
from openmpl_amd.multiview_mpl import MultiView_MPL  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = MultiView_MPL(num_views=4, depth=12, pose_3d_emb_learnable=True).cuda().eval()
detrng.fill_module_(model, seed=7)
poses, rays, centers = detrng.make_inputs(B, 4, 17, seed=3)
mk = lambda xs: [torch.from_numpy(x).cuda() for x in xs]
P, R, Cn = mk(poses), mk(rays), mk(centers)


def run():
    with torch.no_grad():
        return model(P, centers=Cn, rays=R)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


y0 = run().clone()
print("stream launches: %.3f ms" % timeit(run))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        run()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    yg = run()
g.replay()
torch.cuda.synchronize()
print("graph replay:    %.3f ms   bitwise equal: %s" % (timeit(g.replay), bool(torch.equal(yg, y0))))
