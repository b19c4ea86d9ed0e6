"""Paired-column-group instance (NPASS = 2) vs one group per workgroup: bitwise comparison of a whole forward.
    python tools/x3_pair_check.py            (spawns itself with MPL_X3_NOPAIR=1)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ROOT)
    from openmpl_amd import detrng
    from openmpl_amd.multiview_mpl import MultiView_MPL
    out = {}
    for B, V in ((1024, 4), (300, 4), (512, 8)):
        model = MultiView_MPL(num_views=V, depth=2, pose_3d_emb_learnable=True).cuda().eval()
        detrng.fill_module_(model, seed=7)
        poses, rays, centers = detrng.make_inputs(B, V, 17, seed=3)
        mk = lambda xs: [torch.from_numpy(x).cuda() for x in xs]
        with torch.no_grad():
            out["B%d_V%d" % (B, V)] = model(mk(poses), centers=mk(centers), rays=mk(rays)).cpu().numpy()
    np.savez(sys.argv[1], **out)
    sys.exit(0)
res = {}
for tag, env in (("pair", {}), ("nopair", {"MPL_X3_NOPAIR": "1"})):
    f = "/tmp/x3_pair_%s.npz" % tag
    subprocess.check_call([sys.executable, os.path.abspath(__file__), f], env=dict(os.environ, **env))
    res[tag] = np.load(f)
for k in res["pair"].files:
    same = np.array_equal(res["pair"][k].view(np.uint32), res["nopair"][k].view(np.uint32))
    print(k, "BITWISE" if same else "DIFFERENT", float(np.abs(res["pair"][k] - res["nopair"][k]).max()))
