"""Parity at large / ragged batch sizes for both fp32 engines, with the location of the worst pose."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import detrng
from openmpl_amd.multiview_mpl import MultiView_MPL
from oracle import mpl_oracle
for B, V, depth in ((3000, 2, 2), (3008, 2, 2), (4096, 4, 2), (1000, 8, 2)):
    flags = dict(num_views=V, depth=depth, pose_3d_emb_learnable=True)
    m = MultiView_MPL(**flags); detrng.fill_module_(m, seed=5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    p, r, c = detrng.make_inputs(B, V, 17, seed=9)
    P, R, C = ([torch.from_numpy(x) for x in l] for l in (p, r, c))
    ref = mpl_oracle.forward(sd, dict(num_joints=17, embed_dim_ratio=32, num_heads=8, **flags), P, R, C)
    ref64 = mpl_oracle.forward(sd, dict(num_joints=17, embed_dim_ratio=32, num_heads=8, **flags), P, R, C, dtype=torch.float64)
    for prec in ("fp32", "fp32_mfma"):
        m.set_matmul_precision(prec)
        with torch.no_grad():
            out = m([x.cuda() for x in P], rays=[x.cuda() for x in R], centers=[x.cuda() for x in C]).cpu()
        err = (out.double() - ref64).abs().amax(dim=(1, 2)) / ref64.abs().max()
        w = int(err.argmax())
        print("B=%d V=%d %-9s vs fp32 oracle %.2e/%.2e | vs fp64: worst pose %d err %.2e, median %.2e | oracle fp32 vs fp64 %.2e"
              % ((B, V, prec) + mpl_oracle.rel_errors(out, ref) + (w, float(err[w]), float(err.median()),
                 mpl_oracle.rel_errors(ref, ref64)[0])))
