"""Bitwise comparison of GEMM kernel variants (bench-only MPL_GEMM_CFG override) on the proj / fc2 shapes, including
the LayerNorm partial statistics written by the residual epilogue:
    python tools/gemm_bitwise.py 0 7112 7113
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    from openmpl_amd import cabi
    lib = cabi.load()
    out = {}
    g = torch.Generator().manual_seed(1)
    st = lambda: torch.cuda.current_stream().cuda_stream
    for name, M, K, N, epi, ln in [("proj", 4096, 544, 544, 2, False), ("fc2", 4096, 1088, 544, 2, False),
                                   ("ragged", 1000, 544, 544, 2, False), ("ln_gelu", 2048, 544, 544, 1, True),
                                   ("ln_bias", 4000, 544, 544, 0, True)]:
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        R = torch.randn(M, N, generator=g).cuda()
        gam, bet = torch.randn(K, generator=g).cuda(), torch.randn(K, generator=g).cuda()
        Y = torch.zeros(M, N, device="cuda")
        so = torch.zeros(M, max(1, K // 136), 2, device="cuda")   # LayerNorm scratch of the INPUT rows
        rc = lib.mpl_ln_linear(A.data_ptr(), M, K, gam.data_ptr() if ln else None, bet.data_ptr() if ln else None, 1e-6,
                               W.data_ptr(), b.data_ptr(), N, epi, R.data_ptr() if epi == 2 else None, Y.data_ptr(),
                               so.data_ptr() if ln else None, st())
        assert rc == 0, rc
        torch.cuda.synchronize()
        out[name + "_y"] = Y.cpu().numpy()
    # whole forward (exercises the statistics-producing residual epilogue through mpl_block_stack)
    from openmpl_amd import detrng
    from openmpl_amd.multiview_mpl import MultiView_MPL
    for B in (64, 1024):
        model = MultiView_MPL(num_views=4, depth=2, pose_3d_emb_learnable=True).cuda().eval()
        detrng.fill_module_(model, seed=7)
        poses, rays, centers = detrng.make_inputs(B, 4, 17, seed=3)
        mk = lambda xs: [torch.from_numpy(x).cuda() for x in xs]
        with torch.no_grad():
            y = model(mk(poses), centers=mk(centers), rays=mk(rays))
        out["forward_B%d" % B] = y.cpu().numpy()
    np.savez(sys.argv[2], **out)
    sys.exit(0)

cfgs = sys.argv[1:] or ["0", "7112"]
res = {}
for c in cfgs:
    f = "/tmp/gemm_bitwise_%s.npz" % c
    env = dict(os.environ, MPL_GEMM_CFG=c)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", f], env=env)
    res[c] = np.load(f)
ref = res[cfgs[0]]
for c in cfgs[1:]:
    for k in ref.files:
        same = np.array_equal(ref[k].view(np.uint32), res[c][k].view(np.uint32))
        print("cfg %s vs %s  %-10s %s  maxdiff %.3g" % (c, cfgs[0], k, "BITWISE" if same else "DIFFERENT",
                                                       float(np.abs(ref[k] - res[c][k]).max())))
