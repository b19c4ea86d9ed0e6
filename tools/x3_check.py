"""Accuracy and speed of the split-operand (fp32 on bf16 matrix cores) GEMM against the native fp32 MFMA GEMM and an
fp64 reference:   python tools/x3_check.py [D] [M]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import cabi  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 544
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = cabi.load()
g = torch.Generator().manual_seed(0)
st = lambda: torch.cuda.current_stream().cuda_stream


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * 2 ** -0.5))


tot = {"mfma": 0.0, "x3": 0.0}
for name, K, N, epi, ln, m in [("qkv", D, 3 * D, 0, True, M), ("proj", D, D, 2, False, M), ("fc1", D, 2 * D, 1, True, M),
                               ("fc2", 2 * D, D, 2, False, M), ("ragged", D, D, 2, False, 1000)]:
    A = torch.randn(m, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(m, N, generator=g)
    gam, bet = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    a64 = A.double()
    if ln:
        a64 = torch.nn.functional.layer_norm(a64, (K,), gam.double(), bet.double(), 1e-6)
    ref = a64 @ W.double().T + b.double()
    if epi == 1:
        ref = gelu(ref)
    if epi == 2:
        ref = ref + R.double()
    Ad, Wd, bd, Rd, gd, bed = (t.cuda() for t in (A, W, b, R, gam, bet))
    W3 = torch.empty(lib.mpl_split_bf16x3_bytes(N, K), dtype=torch.uint8, device="cuda")
    cabi.check(lib.mpl_split_bf16x3(Wd.data_ptr(), N, K, W3.data_ptr(), st()), "split")
    so = torch.zeros(m, max(1, K // 136), 2, device="cuda")
    res = {}
    for tag in ("mfma", "x3"):
        Y = torch.zeros(m, N, device="cuda")
        if tag == "mfma":
            fn = lambda: lib.mpl_ln_linear(Ad.data_ptr(), m, K, gd.data_ptr() if ln else None, bed.data_ptr() if ln else None,
                                           1e-6, Wd.data_ptr(), bd.data_ptr(), N, epi, Rd.data_ptr() if epi == 2 else None,
                                           Y.data_ptr(), so.data_ptr() if ln else None, st())
        else:
            fn = lambda: lib.mpl_ln_linear_x3(Ad.data_ptr(), m, K, gd.data_ptr() if ln else None,
                                              bed.data_ptr() if ln else None, 1e-6, W3.data_ptr(), bd.data_ptr(), N, epi,
                                              Rd.data_ptr() if epi == 2 else None, Y.data_ptr(), so.data_ptr() if ln else None, st())
        rc = fn()
        assert rc == 0, (tag, name, rc)
        torch.cuda.synchronize()
        y = Y.cpu().double()
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        nrm = ((y - ref).norm() / ref.norm()).item()
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
        res[tag] = (err, nrm, best * 1e3)
        if name != "ragged":
            tot[tag] += best * 1e3
    fl = 2.0 * m * N * K
    print("%-7s M=%d K=%d N=%d | fp32 MFMA: err %.2e / %.2e  %6.1f us %6.1f TF | split x3: err %.2e / %.2e  %6.1f us %6.1f TF" % (
        name, m, K, N, res["mfma"][0], res["mfma"][1], res["mfma"][2], fl / res["mfma"][2] / 1e6,
        res["x3"][0], res["x3"][1], res["x3"][2], fl / res["x3"][2] / 1e6))
print("block total: fp32 MFMA %.1f us, split x3 %.1f us" % (tot["mfma"], tot["x3"]))
