"""Does running two half batches on two HIP streams beat one full batch? (poses are independent)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import detrng
from openmpl_amd.multiview_mpl import MultiView_MPL
B, V = 1024, 4
m = MultiView_MPL(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=12, num_views=V, pose_3d_emb_learnable=True)
detrng.fill_module_(m, seed=11); m = m.cuda().eval()
p, r, c = detrng.make_inputs(B, V, seed=1)
mk = lambda l, a, b: [torch.from_numpy(x[a:b]).cuda() for x in l]
full = (mk(p, 0, B), mk(r, 0, B), mk(c, 0, B))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
with torch.no_grad():
    t1 = timeit(lambda: m(full[0], rays=full[1], centers=full[2]))
    print("1 stream  B=1024: %.3f ms" % t1)
    for parts in (2, 4):
        step = B // parts
        chunks = [(mk(p, i * step, (i + 1) * step), mk(r, i * step, (i + 1) * step), mk(c, i * step, (i + 1) * step)) for i in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        def run():
            cur = torch.cuda.current_stream()
            for s, ch in zip(streams, chunks):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    m(ch[0], rays=ch[1], centers=ch[2])
            for s in streams: cur.wait_stream(s)
        print("%d streams x B=%d: %.3f ms" % (parts, step, timeit(run)))
