"""Determinism soak of the pipelined GEMM kernels: the same forward many times, every output bitwise equal to the first
(a race in the DMA ring / barrier protocol would show up as a rare mismatch).  [SOAK_ONLY=narrow] python tools/soak.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import detrng  # noqa: E402
from openmpl_amd.multiview_mpl import MultiView_MPL  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bad = 0
SHAPES = ((dict(num_views=4, depth=12, pose_3d_emb_learnable=True), 1024),
                 (dict(num_views=4, depth=2, pose_3d_emb_learnable=True, confidence_input_as_third=True, input_rays_as_token=True,
                       multiple_spatial_blocks=True, add_3D_pos_encoding_to_rays=True), 1024),
                 (dict(num_views=8, depth=2, pose_3d_emb_learnable=True), 1000),       # 125 row tiles: pairs, the last one half empty
                 (dict(num_views=4, depth=2, pose_3d_emb_learnable=True), 8190),       # four pairs per team, ragged last tile
                 (dict(num_views=5, depth=2, pose_3d_emb_learnable=True), 333),
                 (dict(num_views=2, depth=12, pose_3d_emb_learnable=True), 1),         # <= 80 token rows: the small-batch engine (sm_stack.hip)
                 (dict(num_views=4, depth=2, pose_3d_emb_learnable=True, confidence_input_as_third=True, input_rays_as_token=True,
                       multiple_spatial_blocks=True, add_3D_pos_encoding_to_rays=True), 4),
                 (dict(num_views=8, depth=12, pose_3d_emb_learnable=True), 2),
                 (dict(num_views=4, depth=12, pose_3d_emb_learnable=True), 4),         # 16 rows: two groups of two sequences
                 (dict(num_views=4, depth=12, pose_3d_emb_learnable=True), 8),         # 32 rows: two full row tiles, side by side
                 (dict(num_views=4, depth=12, pose_3d_emb_learnable=True), 17),        # 68 rows: five groups of sequences, two column tiles per workgroup
                 (dict(num_views=3, depth=2, pose_3d_emb_learnable=True, confidence_as_attention_uncertainty_weight=True), 7),
                 (dict(num_views=6, depth=2, pose_3d_emb_learnable=True, FPT_blocks_view_keypoint_tokens=True), 100),
                 # row-narrow teams: 16-row workgroups in the direct-W form (h2_stackd_kernel), 32-row ones in the ring form
                 (dict(num_views=2, depth=12, pose_3d_emb_learnable=True), 256),
                 (dict(num_views=4, depth=2, pose_3d_emb_learnable=True), 100),
                 (dict(num_views=8, depth=2, pose_3d_emb_learnable=True), 61),
                 (dict(num_views=16, depth=2, pose_3d_emb_learnable=True), 24),
                 (dict(num_views=2, depth=2, pose_3d_emb_learnable=True), 640))
if os.environ.get("SOAK_ONLY") == "narrow":
    SHAPES = SHAPES[-5:]
for flags, B in SHAPES:
    m = MultiView_MPL(**flags).cuda().eval()
    detrng.fill_module_(m, seed=21)
    V = flags["num_views"]
    p, r, c = detrng.make_inputs(B, V, 17, seed=4)
    mk = lambda xs: [torch.from_numpy(x).cuda() for x in xs]
    P, R, C = mk(p), mk(r), mk(c)
    for prec in ("fp32", "fp32_mfma"):
        m.set_matmul_precision(prec)
        with torch.no_grad():
            first = m(P, rays=R, centers=C).clone()
            mism = 0
            for i in range(N):
                out = m(P, rays=R, centers=C)
                if not torch.equal(out, first):
                    mism += 1
        torch.cuda.synchronize()
        print("V=%d B=%d depth=%d %-9s: %d forwards, %d mismatches, finite=%s" % (V, B, flags["depth"], prec, N, mism,
                                                                                bool(torch.isfinite(first).all())))
        bad += mism
sys.exit(1 if bad else 0)
