"""Stack time of the small-batch engine (sm_stack.hip) per launch, from the library's own event pairs.   python tools/sm_time.py [V B]..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi  # noqa: E402
dev = torch.device("cuda", 0)
args = [int(x) for x in sys.argv[1:]] or [2, 1, 2, 8, 4, 8]
for V, B in zip(args[::2], args[1::2]):
    m = build_model(model_flags("chosen", V, 12), dev)
    b = make_batch(B, V, dev, seed=B, step=0)
    with torch.no_grad():
        for _ in range(10):
            m(b[0], rays=b[1], centers=b[2])
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            cabi.profile_start()
            for _ in range(20):
                m(b[0], rays=b[1], centers=b[2])
            torch.cuda.synchronize()
            ts.append(cabi.profile_stop()["gemm"][0] / 20 * 1e3)
    ts.sort()
    print("V=%d B=%d: stack %.1f us per launch (median of 5 x 20; min %.1f max %.1f)%s" % (V, B, ts[2], ts[0], ts[-1], " DEVICE ERROR" if cabi.device_error() else ""), flush=True)
    if cabi.device_error():
        cabi.clear_device_error()
