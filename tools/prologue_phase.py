"""Where the prologue of a stack phase goes (library built with -DH2_DBG=2): entry -> hand-off wait over (the team has arrived)
-> stage 0 .. 2 of this wave landed -> first fragments read (k loop starts).  python tools/prologue_phase.py [M]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openmpl_amd import cabi
from tools._stack_fixture import lib, dev, st, make_block
D, NB = 544, 3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
blks, keep, g = make_block(D)
x = torch.randn(M, D, generator=g).to(dev)
wsb = lib.mpl_block_stack_workspace_bytes(M // 4, 4, D)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
dbg = torch.zeros(8 * 8 * 1024, dtype=torch.int64, device=dev)
names = ["qkv+att", "proj+res", "fc1+gelu", "fc2+res"]
sched = (C.c_uint8 * NB)(*([0] * NB))
def run():
    cabi.check(lib.mpl_block_stack(x.data_ptr(), M // 4, 4, D, 8, blks, sched, NB, ws.data_ptr(), wsb, st()), "stack")
for _ in range(3): run()
torch.cuda.synchronize()
for stop in range(4 * (NB - 1) + 1, 4 * NB + 1):
    lib.mpl_x3_stack_mode(stop << 8)
    dbg.zero_()
    lib.mpl_x3_debug_buffer(dbg.data_ptr()); run(); torch.cuda.synchronize(); lib.mpl_x3_debug_buffer(None)
    t = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t[t[:, 0] > 0]
    ent, loop, chain, land = t[:, 0], t[:, 1], t[:, 5], t[:, 6]
    for role, sel in (("waves 0-3", np.arange(len(t)) % 8 < 4), ("waves 4-7", np.arange(len(t)) % 8 >= 4)):
        u = sel
        print("%-9s %s | entry -> team arrived %5.0f (min %5.0f max %5.0f) | -> own pieces of the first stages landed %5.0f | -> statistics, barrier, first fragments %5.0f | total %5.0f"
              % (names[(stop - 1) & 3], role, (chain - ent)[u].mean(), (chain - ent)[u].min(), (chain - ent)[u].max(), (land - chain)[u].mean(),
                 (loop - land)[u].mean(), (loop - ent)[u].mean()))
lib.mpl_x3_stack_mode(0)
