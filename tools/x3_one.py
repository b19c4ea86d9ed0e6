"""One shape of the split-operand GEMM in a loop (for rocprofv3 --pmc passes): python tools/x3_one.py proj|fc2|fc1|qkv [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import cabi  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "proj"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
D = 544
K, N, epi, ln = {"qkv": (D, 3 * D, 0, True), "proj": (D, D, 2, False), "fc1": (D, 2 * D, 1, True), "fc2": (2 * D, D, 2, False)}[which]
lib = cabi.load()
g = torch.Generator().manual_seed(0)
st = lambda: torch.cuda.current_stream().cuda_stream
A = torch.randn(M, K, generator=g).cuda()
W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
b = torch.randn(N, generator=g).cuda()
R = torch.randn(M, N, generator=g).cuda()
gam, bet = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.1).cuda()
W3 = torch.empty(lib.mpl_split_bf16x3_bytes(N, K), dtype=torch.uint8, device="cuda")
cabi.check(lib.mpl_split_bf16x3(W.data_ptr(), N, K, W3.data_ptr(), st()), "split")
so = torch.zeros(M, max(1, K // 136), 2, device="cuda")
Y = torch.zeros(M, N, device="cuda")
for _ in range(20):
    rc = lib.mpl_ln_linear_x3(A.data_ptr(), M, K, gam.data_ptr() if ln else None, bet.data_ptr() if ln else None, 1e-6,
                              W3.data_ptr(), b.data_ptr(), N, epi, R.data_ptr() if epi == 2 else None, Y.data_ptr(),
                              so.data_ptr() if ln else None, st())
    assert rc == 0
torch.cuda.synchronize()
