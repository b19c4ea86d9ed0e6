"""Per-phase time line of the persistent stack kernel (chain mode) at the headline shape: the stack is stopped after
phase p (mpl_x3_stack_mode) so that the per-wave stamps of mpl_x3_debug_buffer are those of phase p.
[ENGINE=h2|b1] python tools/chain_phase.py [D] [n_blocks] [M] [n_tok]   (library built with -DH2_DBG=1; DBG2=1 with -DH2_DBG=2: the prologue split per wave)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openmpl_amd import cabi

from tools._stack_fixture import lib, dev, st, make_block, ENGINE
D = int(sys.argv[1]) if len(sys.argv) > 1 else 544
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 3
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
NTOK = int(sys.argv[4]) if len(sys.argv) > 4 else 4
blks, keep, g = make_block(D)
x = torch.randn(M, D, generator=g).to(dev)
wsb = lib.mpl_block_stack_workspace_bytes(M // NTOK, NTOK, D)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
dbg = torch.zeros(8 * 8 * 1024, dtype=torch.int64, device=dev)
names = ["qkv+att", "proj+res", "fc1+gelu", "fc2+res"]
sched = (C.c_uint8 * NB)(*([0] * NB))
def run():
    cabi.check(lib.mpl_block_stack(x.data_ptr(), M // NTOK, NTOK, D, 8, blks, sched, NB, ws.data_ptr(), wsb, st()), "stack")
for _ in range(3): run()
torch.cuda.synchronize()
prev_end = None
for stop in range(4 * (NB - 1) - 1, 4 * NB + 1):
    lib.mpl_x3_stack_mode(stop << 8)
    dbg.zero_()
    lib.mpl_x3_debug_buffer(dbg.data_ptr()); run(); torch.cuda.synchronize(); lib.mpl_x3_debug_buffer(None)
    t = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t[t[:, 0] > 0]
    ph = (stop - 1) & 3
    # teams that own >= 2 row tiles run the two-tile stage (h2_stack2_kernel): fc1 is then two one-pass steps of D / 32 stages
    # (the stamps are those of the second), qkv two one-tile steps
    pairs = (M + 63) // 64 > 256 // (D // 136)
    if ENGINE == "h2":
        nst = {0: 3, 1: 1, 2: 1 if pairs else 2, 3: 2}[ph] * (D // 32)
    else:       # bf16: a stage is a pair of k-tiles; every phase runs the pair form
        nst = {0: 3, 1: 1, 2: 2, 3: 1}[ph] * ((D // 32 + 1) // 2) if ph != 3 else (2 * D // 32 + 1) // 2
    ent, loop, epi, sto, end = (t[:, i] for i in range(5))
    print("stop %2d %-9s waves %4d | entry->loop %6.0f (min %6.0f max %6.0f) | k loop %7.0f (%5.0f/stage, %d stages) | epilogue %6.0f | drain %5.0f | total %7.0f | DMA wait/stage %4.0f  bar/stage %4.0f"
          % (stop, names[ph], len(t), (loop - ent).mean(), (loop - ent).min(), (loop - ent).max(), (epi - loop).mean(), (epi - loop).mean() / nst, nst,
             (sto - epi).mean(), (end - sto).mean(), (end - ent).mean(), t[:, 5].mean() / nst, t[:, 6].mean() / nst))
    if os.environ.get("DBG2"):
        # library built with -DH2_DBG=2: columns 5, 6 are the stamps behind the hand-off wait and behind the landing of the first
        # operands; one line per wave index (multiplying and loader waves of the direct-W form differ)
        for w in range(8):
            u = t[np.arange(len(t)) % 8 == w]
            if len(u):
                print("      wave %d: entry->hand-off %6.0f | ->operands landed %6.0f | ->k loop %6.0f | k loop %7.0f | epilogue %6.0f | drain %5.0f"
                      % (w, (u[:, 5] - u[:, 0]).mean(), (u[:, 6] - u[:, 5]).mean(), (u[:, 1] - u[:, 6]).mean(), (u[:, 2] - u[:, 1]).mean(),
                         (u[:, 3] - u[:, 2]).mean(), (u[:, 4] - u[:, 3]).mean()))
        continue
    # by wave role (rows of t are (block, wave): wave = index % 8): per stage DMA wait | lgkm + barrier | MFMA rows | everything else
    for role, sel in (("waves 0-3 (5 tiles, A pieces)", np.arange(len(t)) % 8 < 4), ("waves 4-7 (4 tiles)", np.arange(len(t)) % 8 >= 4)):
        if sel.sum() and t.shape[1] > 7:
            u = t[sel]
            kl = (u[:, 2] - u[:, 1])
            print("      %-30s per stage: DMA wait %4.0f | lgkm+barrier %4.0f | MFMA rows %4.0f | rest (reads, DMA requests, conversion, scalar) %4.0f"
                  % (role, u[:, 5].mean() / nst, u[:, 6].mean() / nst, u[:, 7].mean() / nst, (kl - u[:, 5] - u[:, 6] - u[:, 7]).mean() / nst))
lib.mpl_x3_stack_mode(0)
