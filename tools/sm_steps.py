"""Per-step anatomy of the small-batch engine (library built with -DSM_DBG: tools/build_variants.sh -f sm_stack.hip dbg="-DSM_DBG",
then on the GPU box  bash tools/ab.sh "python tools/sm_steps.py 2 1" -r 1 dbg): shader-clock stamps of every phase of a step for the
workgroups 0 and 50, waves 0 and 1, calibrated against the 100-MHz real-time counter.   python tools/sm_steps.py [V B]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402
dev = torch.device("cuda", 0)
V, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 1)
lib = cabi.load()
m = build_model(model_flags("chosen", V, 12), dev)
b = make_batch(B, V, dev, seed=B, step=0)
with torch.no_grad():
    for _ in range(5):
        m(b[0], rays=b[1], centers=b[2])
    torch.cuda.synchronize()
buf = np.zeros((2, 2, 400, 8), np.uint64)
assert lib.mpl_sm_dbg(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64)
c = t[0, 0, 399]
rate = (c[2] - c[0]) / ((c[3] - c[1]) * 10.0)      # s_memtime ticks per ns
print("clock: %d s_memtime ticks in %d s_memrealtime ticks (100 MHz) = %.1f us; s_memtime rate %.3f GHz" % (c[2] - c[0], c[3] - c[1], (c[3] - c[1]) / 100.0, (c[2] - c[0]) / ((c[3] - c[1]) * 10.0)))
print("calibration inside the launch: 64 dependent fp32 MFMAs = %d / %d s_memtime ticks (application 0 / 6; 2200 on an idle chip)" % (t[0, 0, 398, 0], t[0, 0, 398, 1]))
for wg in range(2):
    print("workgroup %d: waves 0..3 on SIMD %s of CU %s (HW_ID)" % (0 if wg == 0 else 50, [int(x >> 4) & 3 for x in t[wg, 0, 397, :4]], [int(x >> 8) & 15 for x in t[wg, 0, 397, :4]]))
names = ["qkv", "att", "proj", "fc1", "fc2"]
n_steps = 65
for wg in range(2):
    for wave in range(2):
        x = t[wg, wave, :n_steps] / (rate * 1000.0)     # us (s_memtime ticks at the calibrated rate)
        print("workgroup %d wave %d, us per phase (median over the applications; every stamp costs ~0.2 us itself): step | entry->A arrived | LN | weights wait | MFMA | exchange | epilogue | step total (entry to next entry)" % (0 if wg == 0 else 50, wave))
        for ph in range(5):
            rows = [s for s in range(5, n_steps - 5) if s % 5 == ph]
            d = lambda a, c: np.median([x[s, c] - x[s, a] for s in rows if x[s, c] and x[s, a]] or [0])
            nxt = np.median([x[s + 1, 0] - x[s, 0] for s in rows])
            print("  %-4s | %6.2f | %5.2f | %5.2f | %5.2f | %5.2f | %5.2f | %6.2f" % (names[ph], d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), nxt))
        print("  whole stack: %.1f us" % (x[n_steps - 1, 6] - x[0, 0]))
