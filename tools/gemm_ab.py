"""Time the FPT GEMM shapes alone (no LN prologue kernel), for A/B runs of kernel variants:
    MPL_GEMM_VAR=v MPL_GEMM_ABL=a python tools/gemm_ab.py [D] [M]
MPL_GEMM_X3=1 selects the split-operand kernels (x3_gemm.hip) instead of the fp32 MFMA ones.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import cabi  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 544
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = cabi.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
st = lambda: torch.cuda.current_stream().cuda_stream
tag = "var=%s abl=%s" % (os.environ.get("MPL_GEMM_VAR", "default"), os.environ.get("MPL_GEMM_ABL", "0"))
tot_ms, tot_fl = 0.0, 0.0
line = []
SHAPES = [("qkv", D, 3 * D, 0), ("proj", D, D, 2), ("fc1", D, 2 * D, 1), ("fc2", 2 * D, D, 2)]
if os.environ.get("MPL_GEMM_ABL"):
    SHAPES = [("qkv", D, 3 * D, 0), ("proj*", D, D, 0)]
for name, K, N, epi in SHAPES:
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev)
    Y = torch.empty(M, N, device=dev)
    if os.environ.get("MPL_GEMM_X3"):      # split-operand GEMM (fp32 on the bf16 matrix cores)
        W3 = torch.empty(lib.mpl_split_bf16x3_bytes(N, K), dtype=torch.uint8, device=dev)
        cabi.check(lib.mpl_split_bf16x3(W.data_ptr(), N, K, W3.data_ptr(), st()), "split")
        fn = lambda: lib.mpl_ln_linear_x3(A.data_ptr(), M, K, None, None, 0.0, W3.data_ptr(), b.data_ptr(), N, epi,
                                          R.data_ptr() if epi == 2 else None, Y.data_ptr(), None, st())
    else:
        fn = lambda: lib.mpl_ln_linear(A.data_ptr(), M, K, None, None, 0.0, W.data_ptr(), b.data_ptr(), N, epi,
                                       R.data_ptr() if epi == 2 else None, Y.data_ptr(), None, st())
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * K
    tot_ms += best
    tot_fl += fl
    line.append("%s %.1fus %.1fTF" % (name, best * 1e3, fl / best / 1e9))
print("%-18s D=%d M=%d | %s | block %.1fus %.1fTF" % (tag, D, M, " | ".join(line), tot_ms * 1e3, tot_fl / tot_ms / 1e9))
