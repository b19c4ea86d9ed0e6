"""Per-wave phase cycles of the split-operand GEMM k loop (MPL_X3_DBG=1): python tools/x3_phase.py [K]"""
import os
import sys

os.environ["MPL_X3_DBG"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import cabi  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 544
M, N = 4096, 544
lib = cabi.load()
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
A = torch.randn(M, K, generator=g).cuda()
W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
b = torch.randn(N, generator=g).cuda()
R = torch.randn(M, N, generator=g).cuda()
W3 = torch.empty(lib.mpl_split_bf16x3_bytes(N, K), dtype=torch.uint8, device="cuda")
cabi.check(lib.mpl_split_bf16x3(W.data_ptr(), N, K, W3.data_ptr(), st), "split")
Y = torch.zeros(M, N, device="cuda")
dbg = torch.zeros(256 * 8 * 16, device="cuda")
for _ in range(3):
    rc = lib.mpl_ln_linear_x3(A.data_ptr(), M, K, None, None, 1e-6, W3.data_ptr(), b.data_ptr(), N, 2, R.data_ptr(), Y.data_ptr(),
                              dbg.data_ptr(), st)
    assert rc == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.mpl_ln_linear_x3(A.data_ptr(), M, K, None, None, 1e-6, W3.data_ptr(), b.data_ptr(), N, 2, R.data_ptr(), Y.data_ptr(), dbg.data_ptr(), st)
e1.record()
torch.cuda.synchronize()
print('kernel (instrumented) %.1f us per launch' % (e0.elapsed_time(e1) * 50))
d = dbg.view(256, 8, 16).cpu()
n = d[..., 4:5].clamp(min=1)
per = d[..., :4] / n
names = ["batch-0 MFMA + DMA issue", "vmcnt/lgkmcnt waits", "barrier", "reads + split x batch-1"]
for w in (0, 4):
    print("wave %d: prologue %.0f | loop %.0f | epilogue values+stores issued %.0f | stores drained %.0f cycles" % (w, d[:, w, 5].mean(), d[:, w, 6].mean(), d[:, w, 7].mean(), d[:, w, 8].mean()))
print("stages per wave: %d; s_memtime ticks are 100 MHz? -> raw units" % int(d[0, 0, 4]))
for w in range(8):
    print("wave %d (%d tiles): " % (w, 5 if w < 4 else 4) + " | ".join("%s %.0f" % (nm, per[:, w, i].mean()) for i, nm in enumerate(names))
          + " | sum %.0f" % per[:, w].sum(-1).mean())
