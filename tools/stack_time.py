"""Wall time of the block-stack launch against the number of active row-tile teams (same work per workgroup):
a time that grows with the active fraction of the chip is power / clock, not the kernel.  [ENGINE=h2|b1] python tools/stack_time.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmpl_amd import cabi

from tools._stack_fixture import lib, dev, st, make_block, ENGINE
D, NB = 544, 13
blks, keep, g = make_block(D)
sched = (C.c_uint8 * NB)(*([0] * NB))
for M in ([int(a) for a in sys.argv[1:]] or [512, 2048, 3072, 4096, 512]):
    x = (torch.randn(M, D, generator=g) * 0.1).to(dev)
    wsb = lib.mpl_block_stack_workspace_bytes(M // 4, 4, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    run = lambda: cabi.check(lib.mpl_block_stack(x.data_ptr(), M // 4, 4, D, 8, blks, sched, NB, ws.data_ptr(), wsb, st()), "stack")
    for _ in range(30): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(ENGINE, "M = %4d (%3d workgroups): %.3f ms per stack of %d blocks" % (M, M // 64 * 4, e0.elapsed_time(e1) / 50, NB))
