"""Wall time of the block-stack launch against the number of active row-tile teams (same work per workgroup):
a time that grows with the active fraction of the chip is power / clock, not the kernel.  python tools/stack_time.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmpl_amd import cabi

lib = cabi.load()
D, NB, dev = 544, 13, "cuda"
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
sched = (C.c_uint8 * NB)(*([0] * NB))
for M in (512, 2048, 3072, 4096, 512):
    x = (torch.randn(M, D, generator=g) * 0.1).to(dev)
    wsb = lib.mpl_block_stack_workspace_bytes(M // 4, 4, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    run = lambda: cabi.check(lib.mpl_block_stack(x.data_ptr(), M // 4, 4, D, 8, blks, sched, NB, ws.data_ptr(), wsb, st()), "stack")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print("M = %4d (%3d workgroups): %.3f ms per stack of %d blocks" % (M, M // 64 * 4, e0.elapsed_time(e1) / 50, NB))
