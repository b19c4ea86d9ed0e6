"""In-kernel phase timing of the GEMM kernels (MPL_GEMM_ABL=4 instrumentation, shader-clock cycles):
per stage: DMA wait / barrier / DMA issue / fragment+MFMA; per kernel: prologue, loop, epilogue; dispatch skew."""
import os, sys
os.environ["MPL_GEMM_ABL"] = "4"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import cabi
lib = cabi.load()
M = 4096
D = int(sys.argv[1]) if len(sys.argv) > 1 else 544
g = torch.Generator().manual_seed(0)
st = lambda: torch.cuda.current_stream().cuda_stream
for name, K, N, epi in [("qkv", D, 3 * D, 0), ("proj", D, D, 2), ("fc1", D, 2 * D, 1), ("fc2", 2 * D, D, 2)]:
    A = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    Y = torch.zeros(M, N, device="cuda"); R = torch.randn(M, N, generator=g).cuda()
    dbg = torch.zeros(1 << 20, device="cuda")
    for _ in range(3):
        lib.mpl_ln_linear(A.data_ptr(), M, K, None, None, 0.0, W.data_ptr(), b.data_ptr(), N, epi,
                          R.data_ptr() if epi == 2 else None, Y.data_ptr(), dbg.data_ptr(), st())
    torch.cuda.synchronize()
    nw = 12 if (N == 3 * D) else 4
    nwg = (M // 64) * (N // (136 * (3 if nw == 12 else 1)))
    r = dbg[: nwg * nw * 12].reshape(nwg * nw, 12).cpu()
    T = r[:, 5].mean().item()
    m = r.mean(0)
    t0 = r[:, 8]
    skew = ((t0 - t0.min()) % (1 << 24)).float() / 100.0     # us (100 MHz)
    print("%-5s WGs=%d stages=%d | per stage: wait_dma %.0f barrier %.0f dma_issue %.0f frag+mfma %.0f (ideal %d) | "
          "prologue %.0f loop %.0f epilogue %.0f total %.0f cycles = %.1f us @2.4GHz | start skew: mean %.1f max %.1f us"
          % (name, nwg, T, m[0] / T, m[1] / T, m[2] / T, m[3] / T, 72 * 32 * (3 if nw == 12 else 1), m[6], m[4], m[7], m[9],
             m[9] / 2400.0, skew.mean().item(), skew.max().item()))
