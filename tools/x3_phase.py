"""Where a split-operand GEMM launch spends its time: per-wave shader-clock stamps (mpl_x3_debug_buffer) of the four
GEMMs of one FPT block at the headline shape (M = 4096, D = 544).   python tools/x3_phase.py [D]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openmpl_amd import cabi

lib = cabi.load()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 544
M, dev = 4096, "cuda"
st = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
def operand(N, K, ln):
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
    gam = (torch.rand(K, generator=g) + 0.5).to(dev); bet = (torch.randn(K, generator=g) * 0.1).to(dev)
    o = torch.empty(lib.mpl_split_bf16x3_bytes(N, K), dtype=torch.uint8, device=dev)
    cabi.check(lib.mpl_split_bf16x3(W.data_ptr(), b.data_ptr(), gam.data_ptr() if ln else None, bet.data_ptr() if ln else None, N, K, o.data_ptr(), st()), "split")
    return o
blk = cabi.BlockWeights()
keep = [operand(3 * D, D, True), operand(D, D, False), operand(2 * D, D, True), operand(D, 2 * D, False)]
blk.qkv_w3, blk.proj_w3, blk.fc1_w3, blk.fc2_w3 = (k.data_ptr() for k in keep)
blks = (cabi.BlockWeights * 1)(blk)
x = torch.randn(M, D, generator=g).to(dev)
wsb = lib.mpl_block_stack_workspace_bytes(M // 4, 4, D)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
dbg = torch.zeros(8 * 8 * 1024, dtype=torch.int64, device=dev)
names = ["LN1+qkv+attention", "proj+residual", "LN2+fc1+GELU", "fc2+residual"]
sched = (C.c_uint8 * 1)(0)
def run(n_apps=1):
    cabi.check(lib.mpl_block_stack(x.data_ptr(), M // 4, 4, D, 8, blks, sched, n_apps, ws.data_ptr(), wsb, st()), "stack")
for _ in range(3): run()
torch.cuda.synchronize()
# one launch at a time cannot be isolated through mpl_block_stack: every launch overwrites the stamps, so the buffer holds
# the LAST GEMM (fc2) after a full block; the others are reached by stacks truncated with an invalid later operand
for upto, name in ((4, names[3]),):
    lib.mpl_x3_debug_buffer(dbg.data_ptr()); run(); torch.cuda.synchronize(); lib.mpl_x3_debug_buffer(None)
    t = dbg.cpu().numpy().reshape(-1, 8)[:256 * 8, :8].astype(np.float64)
    t0 = t[:, 0].min()
    print("%-20s waves %d  entry spread %.0f | prologue %.0f | k loop %.0f (%.0f per stage) | epilogue to stores issued %.0f | drain %.0f | total %.0f  (shader-clock ticks, mean over waves; 100 MHz ticks if constant clock)"
          % (name, len(t), (t[:, 0] - t0).max(), (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 2] - t[:, 1]).mean() / (2 * D // 32),
             (t[:, 3] - t[:, 2]).mean(), (t[:, 4] - t[:, 3]).mean(), (t[:, 4].max() - t0)))
    nst = 2 * D // 32
    for half, nm in ((slice(0, None, 8), "wave 0 (5 tiles)"), (slice(4, None, 8), "wave 4 (4 tiles)")):
        print("   %s: per stage  DMA wait %.0f | lgkm + barrier %.0f | rest (issue MFMA + reads + DMA) %.0f"
              % (nm, t[half, 5].mean() / nst, t[half, 6].mean() / nst, ((t[half, 2] - t[half, 1]) - t[half, 5] - t[half, 6]).mean() / nst))
